"""Independent torch-CPU restatement of the VNect graph (src/vnect_model.py:25-217) in float64.

Written separately from oracle/vnect_net.c (NCHW, explicit F.pad, F.conv_transpose2d) so that
agreement between the two is evidence for both; TensorFlow 1.x itself cannot run here.
Test helper only.
"""
import torch
import torch.nn.functional as F


def _w(weights, name, dtype):
    return torch.from_numpy(weights[name]).to(dtype)


def _conv(x, weights, scope, k, stride=1, relu=True, dtype=torch.float64):
    w = _w(weights, scope + "/weights", dtype).permute(3, 2, 0, 1)  # HWIO -> OIHW
    b = _w(weights, scope + "/biases", dtype)
    if k > 1:  # TF SAME: total = max((ceil(n/s)-1)*s + k - n, 0); before = total // 2
        n = x.shape[2]
        total = max((-(-n // stride) - 1) * stride + k - n, 0)
        lo, hi = total // 2, total - total // 2
        x = F.pad(x, (lo, hi, lo, hi))
    y = F.conv2d(x, w, b, stride=stride)
    return F.relu(y) if relu else y


def forward(weights, batch_nhwc, paper_res2c=False, dtype=torch.float64, taps=None):
    """batch (S,368,368,3) numpy f32 -> (S,46,46,84) torch tensor (NHWC); taps: dict filled with NHWC layer outputs."""
    x = torch.from_numpy(batch_nhwc).to(dtype).permute(0, 3, 1, 2)

    def keep(name, t):
        if taps is not None:
            taps[name] = t.permute(0, 2, 3, 1).contiguous()
        return t

    c = lambda t, scope, k, **kw: keep(scope, _conv(t, weights, scope, k, dtype=dtype, **kw))
    conv1 = c(x, "conv1", 7, stride=2)
    pool1 = keep("pool1", F.max_pool2d(F.pad(conv1, (0, 1, 0, 1), value=float("-inf")), 3, 2))

    def proj(t, p, stride):
        s = c(t, p + "_branch1", 1, stride=stride, relu=False)
        a = c(t, p + "_branch2a", 1, stride=stride)
        b = c(a, p + "_branch2b", 3)
        return keep(p, F.relu(_conv(b, weights, p + "_branch2c", 1, relu=False, dtype=dtype) + s))

    def ident(t, p):
        a = c(t, p + "_branch2a", 1)
        b = c(a, p + "_branch2b", 3)
        return keep(p, F.relu(_conv(b, weights, p + "_branch2c", 1, relu=False, dtype=dtype) + t)), a

    r2a = proj(pool1, "res2a", 1)
    r2b, r2b_2a = ident(r2a, "res2b")
    a = c(r2b, "res2c_branch2a", 1) if paper_res2c else r2b_2a  # vnect_model.py:54-57 wiring quirk
    b = c(a, "res2c_branch2b", 3)
    r = keep("res2c", F.relu(_conv(b, weights, "res2c_branch2c", 1, relu=False, dtype=dtype) + r2b))
    r = proj(r, "res3a", 2)
    for p in "bcd":
        r, _ = ident(r, "res3" + p)
    r = proj(r, "res4a", 2)
    for p in "bcdef":
        r, _ = ident(r, "res4" + p)
    a = c(r, "res5a_branch2a_new", 1)
    b = c(a, "res5a_branch2b_new", 3)
    s = c(r, "res5a_branch1_new", 1, relu=False)
    r = keep("res5a", F.relu(_conv(b, weights, "res5a_branch2c_new", 1, relu=False, dtype=dtype) + s))
    a = c(r, "res5b_branch2a_new", 1)
    b = c(a, "res5b_branch2b_new", 3)
    r = c(b, "res5b_branch2c_new", 1)

    def deconv(t, scope):
        w = _w(weights, scope + "/kernel", dtype).permute(3, 2, 0, 1)  # (kh,kw,Cout,Cin) -> (Cin,Cout,kh,kw)
        return keep(scope, F.conv_transpose2d(t, w, stride=2, padding=1))

    d1 = deconv(r, "res5c_branch1a")
    d2 = deconv(r, "res5c_branch2a")
    g, be, mu, va = (_w(weights, "bn5c_branch2a/" + n, dtype).view(1, -1, 1, 1)
                     for n in ("gamma", "beta", "moving_mean", "moving_variance"))
    bn = F.relu((d2 - mu) * (g / torch.sqrt(va + 0.001)) + be)
    dx, dy, dz = d1[:, 0:21], d1[:, 21:42], d1[:, 42:63]
    bone = torch.sqrt(dx * dx + dy * dy + dz * dz)
    feat = keep("res5c_branch2a_feat", torch.cat([bn, dx, dy, dz, bone], dim=1))
    h = c(feat, "res5c_branch2b", 3)
    wk = _w(weights, "res5c_branch2c/kernel", dtype).permute(3, 2, 0, 1)
    out = keep("res5c_branch2c", F.conv2d(h, wk))
    return out.permute(0, 2, 3, 1).contiguous()
