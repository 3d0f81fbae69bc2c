"""A VNect weight set whose right answer is known: planted heat-map peaks THROUGH the real network (test input, not a special path).

No trained weights ship with the reference (models/*/README.md), and the heat-maps of random weights are noise: they have no maximum a
test could call "the joint".  This module builds weights in the reference schema (src/vnect_model.py:219-236) for which they do:

* three PASS channels (indices 0, 1, 2 at every width) carry a blurred, sub-sampled copy of the frame's B, G and R intensity through the
  whole graph of src/vnect_model.py:27-217 -- conv1 as a 7x7 blur whose bias turns black into 0, pool1, the `branch1` 1x1 layers
  (stride 2 at res3a / res4a: even pixels) and the identity shortcuts of the residual blocks, res5b's three layers, the transposed conv
  res5c_branch2a as `out[2i] = in[i], out[2i+1] = 0.35 in[i]` (+ BN as the identity), the centre tap of res5c_branch2b -- isolated from
  every other channel (no weight connects a pass channel with a random one);
* res5c_branch2c's heat-map column j is pass channel j % 3 with weight `peak * GAIN[j // 3]` plus `noise` times the seeded random
  column over the random channels: heat-map j = a sharp bump on the brightest blob of colour j % 3 over a noise floor that carries the
  random net's real rounding error; the 63 location-map columns stay random;
* every other weight is the seeded synthetic one, so the random part of the net runs at its usual amplitudes.

`frame()` paints one bright Gaussian blob per colour (and a dimmer distractor each) over a dim texture, on lattice points where a blob
falls on ONE cell of the 23 x 23 stage at every scale of the pyramid; `expected()` says where joint j must come out: ON the blob, to the
pixel -- a geometric known answer for conv1's and pool1's asymmetric SAME padding, the even-pixel sampling of the stride-2 1x1 layers,
the transposed conv's `out[2i - 1 + ky]` alignment, the [heat | x | y | z] channel split, the crop offsets of the multi-scale merge and the
un-mapping, none of which is taken from another restatement.  Every layer runs the same kernels as with any other weights.
"""
import numpy as np

from vnect_amd.weights import synthetic_weights, uniform01

P = 3                       # pass channels: B, G, R
GAIN = [1.0, 0.97, 0.94, 0.91, 0.88, 0.85, 0.82]                     # joint j's heat-map column weighs its colour by peak * GAIN[j // 3]
BLUR = np.array([1, 2, 3, 4, 3, 2, 1], np.float64) / 16.0          # conv1's 7 taps per axis
DECONV = np.array([0.0, 1.0, 0.35, 0.0])                           # out[2i] = in[i], out[2i + 1] = 0.35 in[i]: a unique top cell


def weights(peak=8.0, noise=1.0, base=None):
    w = {k: np.array(v, np.float32, copy=True) for k, v in (base if base is not None else synthetic_weights()).items()}
    for name, a in w.items():
        scope, leaf = name.split("/")
        if scope == "conv1":
            continue
        if leaf == "weights":               # (kh, kw, Cin, Cout): cut every connection between pass and random channels
            a[:, :, :P, :] = 0
            a[:, :, :, :P] = 0
        elif leaf == "biases":
            a[:P] = 0
    # conv1: channel c = blur of colour c, bias so that black gives 0 (the graph's input is x / 255 - 0.4)
    k = w["conv1/weights"]
    k[:, :, :, :P] = 0
    for c in range(P):
        k[:, :, c, c] = np.outer(BLUR, BLUR)
    w["conv1/biases"][:P] = 0.4
    for scope in ("res2a_branch1", "res3a_branch1", "res4a_branch1", "res5a_branch1_new", "res5b_branch2a_new", "res5b_branch2c_new"):
        for c in range(P):
            w[scope + "/weights"][0, 0, c, c] = 1.0
    for c in range(P):
        w["res5b_branch2b_new/weights"][1, 1, c, c] = 1.0
    for name in ("res5c_branch1a/kernel", "res5c_branch2a/kernel"):    # (kh, kw, Cout, Cin)
        w[name][:, :, :, :P] = 0
    d = w["res5c_branch2a/kernel"]
    d[:, :, :P, :] = 0
    for c in range(P):
        d[:, :, c, c] = np.outer(DECONV, DECONV)
    w["bn5c_branch2a/gamma"][:P] = 1.0
    w["bn5c_branch2a/beta"][:P] = 0.0
    w["bn5c_branch2a/moving_mean"][:P] = 0.0
    w["bn5c_branch2a/moving_variance"][:P] = 1.0
    k = w["res5c_branch2b/weights"]          # (3, 3, 212, 128): joint j <- colour j % 3 through tap j // 3
    k[:, :, :, :21] = 0
    w["res5c_branch2b/biases"][:21] = 0
    for j in range(21):
        k[1, 1, j % 3, j] = 1.0
    k = w["res5c_branch2c/kernel"]           # (1, 1, 128, 84): columns [0, 21) are the heat-maps
    k[0, 0, :21, :] = 0
    k[0, 0, 21:, :21] *= noise
    for j in range(21):
        k[0, 0, j, j] = peak * GAIN[j // 3]
    return w


LATTICE = [116, 196, 276]   # box-pixel coordinates where a blob sits on ONE cell of the 23 x 23 stage at all of the scales 1.0, 0.8, 0.6:
                            # 184 + 12 + 80 k maps to 184 + 0.8 * 12 + 64 k and 184 + 0.6 * 12 + 48 k -- the same phase of the 16-pixel
                            # grid (within 3 pixels) at every scale, so the three scales' peaks fall into the same heat-map cell


def frame(seed, H=368, W=368, sigma=6.0, texture=0.2):
    """uint8 BGR frame, dim texture, with one bright Gaussian blob per colour and one at 55 % of its height, on six of the nine LATTICE points
    (given in the 368-box's coordinates and mapped back to the frame).  Returns (frame, centres[colour] = (row, col) in frame pixels)."""
    u = uniform01(seed, 16)
    order = np.argsort(u[:9])                     # a seeded permutation of the nine lattice points
    scaler = 368.0 / max(H, W)
    sh, sw = int(round(H * scaler)), int(round(W * scaler))
    offy, offx = 184 - sh // 2, 184 - sw // 2     # src/utils.py:98-103 (the squarified image is centred)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    # a dim smooth texture under the blobs (an 8 x 8 random grid, bilinearly upsampled, at most `texture` of full scale): the random part
    # of the net then sees a different picture in every frame
    g = uniform01(seed + 77, 9 * 9 * 3).reshape(9, 9, 3).astype(np.float64)
    ys, xs = np.linspace(0, 8, H, endpoint=False), np.linspace(0, 8, W, endpoint=False)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    img = ((g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx) * (1 - fy) + (g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx) * fy) * 255.0 * texture
    centres = []
    for i in range(6):
        by, bx = LATTICE[order[i] // 3], LATTICE[order[i] % 3]
        # OpenCV's pixel-centre convention: box pixel b <-> frame coordinate (b - off + 0.5) / scaler - 0.5
        r, c = (by - offy + 0.5) / scaler - 0.5, (bx - offx + 0.5) / scaler - 0.5
        if not (0 <= r < H and 0 <= c < W):
            continue                              # letter-boxed frames do not contain every lattice point
        col = [k for k in range(3) if sum(1 for cc in centres if cc[0] == k) < 1]
        amp, k = (255.0, col[0]) if col else (140.0, i % 3)
        if col:
            centres.append((k, r, c))
        sg = sigma / scaler
        img[:, :, k] += amp * np.exp(-((yy - r) ** 2 + (xx - c) ** 2) / (2 * sg * sg))
    cen = {k: (r, c) for k, r, c in centres}
    return np.clip(img, 0, 255).astype(np.uint8), [cen[k] for k in range(3)]


def expected(centres):
    """joints_2d ([row, col] in frame pixels) the planted weights must produce for a frame from frame(): joint j sits on the bright blob
    of colour j % 3."""
    return np.array([centres[j % 3] for j in range(21)], np.float64)


def scene(H, W, blobs, sigma, seed=0, texture=0.15):
    """uint8 BGR frame (H, W, 3): the dim texture of frame() with Gaussian blobs `blobs` = [(row, col, colour 0..2, amplitude 0..255)] of
    width `sigma` (frame pixels) anywhere -- the tracking loop's crops do not keep a blob on the lattice of frame(), so peaks through a
    crop are broader and the known answer is "within a heat-map cell or two of the blob", not "to the pixel"."""
    g = uniform01(seed + 77, 9 * 9 * 3).reshape(9, 9, 3).astype(np.float64)
    ys, xs = np.linspace(0, 8, H, endpoint=False), np.linspace(0, 8, W, endpoint=False)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    img = ((g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx) * (1 - fy) + (g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx) * fy) * 255.0 * texture
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for r, c, k, amp in blobs:
        img[:, :, k] += amp * np.exp(-((yy - r) ** 2 + (xx - c) ** 2) / (2.0 * sigma * sigma))
    return np.clip(img, 0, 255).astype(np.uint8)
