// rt_plan.cpp -- from the weight dictionary to the launch plan: packed device layouts, layers and their tiles / fused forms, the
// activation arena, the resize tables (the arithmetic itself is in hostplan.h, HIP-free and sanitizer-tested on the CPU).
// The graph follows /root/reference/src/vnect_model.py:25-217; the weight names are its schema (:219-236).
#include "runtime.h"

namespace vnect {
namespace rt {

using plan::from_bf16;
using plan::to_bf16;
// packed weights: fp32 as is, or converted to bf16 (the device pointer is typed float* either way)
static int upload_weights(vnect_handle* h, float** dst, const std::vector<float>& v)
{
    if (!h->bf16) return upload(h, dst, v);
    std::vector<uint16_t> b(v.size());
    for (size_t i = 0; i < v.size(); i++) b[i] = to_bf16(v[i]);
    uint16_t* p = nullptr;
    int rc = upload(h, &p, b);
    *dst = (float*)p;
    return rc;
}

// Weights of a layer in the layout its launch reads: the split-product form (conv.hip, X3) for 64x64-tile layers of a VNECT_FP32_SPLIT
// handle -- three bf16 planes per row and 32-element K chunk -- else upload_weights.  Sets a.x3.
static int upload_layer_weights(vnect_handle* h, Layer& L, const std::vector<float>& wp)
{
    ConvArgs& a = L.a;
    a.x3 = h->x3 && !h->bf16 && L.BM == 64 && L.BN * L.KG == 64 && !a.pixmode && a.K % 32 == 0 && !getenv("VNECT_NO_X3");
    if (!a.x3) return upload_weights(h, &L.w, wp);
    std::vector<uint16_t> pl;
    plan::pack_split3(wp, std::max(a.nphase, 1), a.Npad, a.K, pl);  // (the transposed conv: 4 phases)
    uint16_t* p = nullptr;
    int rc = upload(h, &p, pl);
    L.w = (float*)p;
    return rc;
}

// ---- tables: the arithmetic is in hostplan.h (HIP-free, sanitizer-tested on the CPU box); here they are built and uploaded ----
int build_scale_tables(vnect_handle* h)
{
    ScaleTabs st;
    memset(&st, 0, sizeof st);
    MergeGeo mg;
    memset(&mg, 0, sizeof mg);
    st.S = mg.S = h->S;
    plan::fill_lut(st.lut);
    for (int i = 0; i < h->S; i++) {
        const double s = h->cfg.scales[i];
        if (const char* why = plan::build_scale_tab(s, &st, i)) return fail(h, VNECT_E_ARG, why);
        if (const char* why = plan::build_merge_geo(s, &mg, i)) return fail(h, VNECT_E_ARG, why);
    }
    HIPCK(h, hipMemcpy(h->d_stabs, &st, sizeof st, hipMemcpyHostToDevice));
    h->mgeo = mg;
    h->stabs_host = st;
    // the stem's from-the-frame form: do all tiles' frame rectangles still fit its LDS scratch at these scales?  (lanes share the tables)
    if (h->stem_mode == 2) h->stem_frame_ok = plan::stem_frame_fits(st, h->stem.S, h->stem.scale_base, h->stem.groups, h->stem.row0, h->bf16);
    for (vnect_handle* tw : h->twins) tw->stabs_host = st, tw->mgeo = mg, tw->stem_frame_ok = h->stem_frame_ok;
    return VNECT_OK;
}

int build_up_table(vnect_handle* h)
{
    std::vector<UpTab> u(1);
    // the arg-max kernels walk the rows of the x8 upsample by (segment, phase) and compute the entries themselves: plan::build_up_tab
    // builds the full table with the same formulas and checks that it really has that structure
    if (!plan::build_up_tab(&u[0])) return fail(h, VNECT_E_STATE, "internal: x8 upsample table does not have the segment/phase structure");
    return VNECT_OK;
}

// utils.img_scale_squarify + img_padding geometry for an (H,W) frame
int squarify_params(vnect_handle* h, int H, int W, FrameParams* fp)
{
    if (h->sq_H != H || h->sq_W != W) {
        FrameParams c;
        if (const char* why = plan::squarify(H, W, &c)) return fail(h, VNECT_E_ARG, why);
        h->sq_cache = c, h->sq_H = H, h->sq_W = W;
    }
    *fp = h->sq_cache;
    return VNECT_OK;
}

// ---- network construction --------------------------------------------------------------------------
int add_tensor(vnect_handle* h, const std::string& name, int S, int H, int W, int C, int Cs, bool force_f32)
{
    Tensor t;
    t.name = name, t.S = S, t.H = H, t.W = W, t.C = C, t.Cs = Cs;
    t.esz = (h->bf16 && !force_f32) ? 2 : 4;
    h->tensors.push_back(t);
    h->tensor_by_name[name] = (int)h->tensors.size() - 1;
    return (int)h->tensors.size() - 1;
}

static const HostArray* get_w(vnect_handle* h, const std::string& name, std::vector<int64_t> shape)
{
    auto it = h->weights.find(name);
    if (it == h->weights.end()) {
        h->err = "missing weight array " + name;
        return nullptr;
    }
    if (it->second.shape != shape) {
        h->err = "weight " + name + " has the wrong shape";
        return nullptr;
    }
    return &it->second;
}

using plan::round_up;
using plan::same_pad;

// Tile and K-split of a layer: plan::choose_tile (hostplan.h) with the tuning overrides VNECT_FORCE_TILE / VNECT_PLAN.
static void choose_tile(Layer& L, long long npix, bool allow96 = false)
{
    (void)npix;
    // (allow96: the transposed conv's three-accumulator shape, fp32 instruction path only -- not on a split-product handle, whose 64x64
    // split-product loop is the faster one for this layer; VNECT_NO_DECONV96=1 restores the 64x64 plan for A/B runs)
    const plan::TileChoice c = plan::choose_tile(L.a.M, L.Nreal, L.a.ntaps, L.a.cpt, L.a.K, L.a.nphase, L.a.bf16 != 0, L.name,
                                                 getenv("VNECT_FORCE_TILE"), getenv("VNECT_PLAN"), allow96 && conv_deconv96_available() && !getenv("VNECT_NO_DECONV96"),
                                                 conv_cu_count());
    L.BM = c.BM, L.BN = c.BN, L.KG = c.KG, L.a.ksplit = c.ks;
}

struct ConvSpec {
    std::string scope, out_name;
    int in = -1, resid = -1;
    int k = 1, stride = 1, cout = 0;
    bool relu = false;
};

// tc.layers.conv2d scope -> Layer (weights HWIO + bias); returns the output tensor index or -1
static int add_conv(vnect_handle* h, const ConvSpec& sp)
{
    const Tensor tin = h->tensors[sp.in];
    const bool conv1 = sp.k == 7;
    const int cin = tin.C;
    const HostArray* W = get_w(h, sp.scope + "/weights", {sp.k, sp.k, cin, sp.cout});
    const HostArray* B = W ? get_w(h, sp.scope + "/biases", {sp.cout}) : nullptr;
    if (!W || !B) return -1;
    int ho, wo, pt = 0, pl = 0;
    if (sp.k == 1) ho = (tin.H - 1) / sp.stride + 1, wo = (tin.W - 1) / sp.stride + 1;  // VALID
    else same_pad(tin.H, sp.k, sp.stride, &ho, &pt), same_pad(tin.W, sp.k, sp.stride, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sp.scope, L.in = sp.in, L.resid = sp.resid;
    const bool final_maps = sp.scope == "res5c_branch2c";  // feeds the f64 post-processing: stays fp32
    L.out = add_tensor(h, sp.out_name, tin.S, ho, wo, sp.cout, sp.cout, final_maps);
    ConvArgs& a = L.a;
    a.out_f32 = final_maps;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo;
    a.stride = sp.stride;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = sp.cout, a.ldr = sp.cout;
    a.relu_cols = sp.relu ? sp.cout : 0;
    a.Nvalid = sp.cout;
    L.Nreal = sp.cout;
    const int EPR = h->bf16 ? 64 : 32;  // K-elements per chunk
    a.bf16 = h->bf16;
    int cp;  // channels per tap in the packed K
    if (conv1) {
        // fp32: K = 7 rows x (8 pixels x 4 channels); bf16: K = 4 row pairs x (2 rows x 8 pixels x 4 channels).
        // Pixel 7, channel 3 (and row 7 in bf16) carry zero weights.
        a.pixmode = 1, a.cpt = 1, cp = 32;
        a.ntaps = h->bf16 ? 4 : 7;
        for (int t = 0; t < a.ntaps; t++) L.dy[t] = (int)((h->bf16 ? 2 * t : t) - pt), L.dx[t] = (int)(-pl);
    } else {
        cp = round_up(tin.Cs, EPR);
        if (cp != tin.Cs) {
            h->err = "internal: input channel stride not a multiple of the chunk at " + sp.scope;
            return -1;
        }
        a.pixmode = 0, a.ntaps = sp.k * sp.k, a.cpt = cp / EPR;
        for (int ky = 0; ky < sp.k; ky++)
            for (int kx = 0; kx < sp.k; kx++) L.dy[ky * sp.k + kx] = (int)(ky - pt), L.dx[ky * sp.k + kx] = (int)(kx - pl);
    }
    a.K = a.ntaps * a.cpt * EPR;
    L.Kreal = sp.k * sp.k * cin;
    L.flops = 2.0 * a.M * (double)L.Kreal * sp.cout;
    choose_tile(L, (long long)a.M);
    a.Npad = round_up(sp.cout, L.BN);
    std::vector<float> wp((size_t)a.Npad * a.K, 0.f), bp(a.Npad, 0.f);
    plan::pack_conv(W->d.data(), sp.k, cin, sp.cout, cp, conv1, h->bf16, a.K, 0, wp);
    for (int n = 0; n < sp.cout; n++) bp[n] = B->d[n];
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp)) return -1;
    h->layers.push_back(L);
    return L.out;
}

// Two convs of ONE input with one kernel size and stride as ONE launch: weights concatenated along N ([a | b]), two output
// tensors, ReLU per column block.  Used for (i) branch2a (ReLU) + branch1 (none), the two 1x1 convs at the head of a projection
// block (vnect_model.py:32-35,64-67,106-109,168-175), and (ii) res2b_branch2b + res2c_branch2b, two 3x3 convs that both read
// res2b_branch2a in the reference's wiring (vnect_model.py:50,56).  Returns the first tensor, *second gets the other one.
//
// Head split (round 5): a pair whose 64x64 tiles need one round over the CUs more than their matrix work does -- res5a_branch2a_new +
// res5a_branch1_new at three scales: 600 tiles = 2.34 per CU = THREE block K loops per SIMD -- runs the first `head` channels of layer a
// as a launch of their own in a K-group shape (64x32x2: half a K loop per SIMD) and the rest as the pair: 500 tiles = two rounds, 2.5 K
// loops in all; plan::pair_head_cols weighs that against the extra launch.  Both launches write the same tensors (the pair from channel
// `head` on: Layer::out_col0); the pair's channels keep their tiles and K order (bit-identical to the single launch), the head's are
// summed by two K groups like every 64x32x2 layer (equal to fp32 rounding).  VNECT_NO_HEAD_SPLIT=1: A/B runs and the parity test.
static int add_conv_pair(vnect_handle* h, const std::string& sa, int cout_a, const std::string& sb, int cout_b, int in,
                  int stride, int* second, int k = 1, bool relu_b = false)
{
    const Tensor tin = h->tensors[in];
    const int cin = tin.C;
    const HostArray* Wa = get_w(h, sa + "/weights", {k, k, cin, cout_a});
    const HostArray* Ba = Wa ? get_w(h, sa + "/biases", {cout_a}) : nullptr;
    const HostArray* Wb = Ba ? get_w(h, sb + "/weights", {k, k, cin, cout_b}) : nullptr;
    const HostArray* Bb = Wb ? get_w(h, sb + "/biases", {cout_b}) : nullptr;
    if (!Bb) return -1;
    const int EPR = h->bf16 ? 64 : 32;
    if (cout_a % 64 || tin.Cs % EPR || (k != 1 && (k != 3 || stride != 1))) {
        h->err = "internal: paired conv needs a 64-aligned split, whole K chunks and 1x1 or 3x3 stride 1";
        return -1;
    }
    int ho = (tin.H - 1) / stride + 1, wo = (tin.W - 1) / stride + 1, pt = 0, pl = 0;  // 1x1: VALID
    if (k == 3) same_pad(tin.H, 3, 1, &ho, &pt), same_pad(tin.W, 3, 1, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sa + "+" + (k == 1 ? sb.substr(sb.find('_') + 1) : sb), L.in = in;
    L.out = add_tensor(h, sa, tin.S, ho, wo, cout_a, cout_a);
    L.out2 = add_tensor(h, sb, tin.S, ho, wo, cout_b, cout_b);
    ConvArgs& a = L.a;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo, a.stride = stride;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = cout_a, a.ldc2 = cout_b, a.split_n = cout_a, a.ldr = 0;
    a.relu_cols = relu_b ? cout_a + cout_b : cout_a;
    a.ntaps = k * k, a.cpt = tin.Cs / EPR, a.K = k * k * tin.Cs;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) L.dy[ky * k + kx] = ky - pt, L.dx[ky * k + kx] = kx - pl;
    a.bf16 = h->bf16;
    L.Nreal = cout_a + cout_b, L.Kreal = k * k * cin;
    a.Nvalid = L.Nreal;
    L.flops = 2.0 * a.M * (double)L.Kreal * L.Nreal;
    choose_tile(L, (long long)a.M);
    if (L.BN != 64 || L.a.ksplit != 1 || L.KG != 1) L.BM = 64, L.BN = 64, L.KG = 1, L.a.ksplit = 1;  // the column split relies on 64-wide tiles, no slabs
    a.Npad = round_up(L.Nreal, 64);
    std::vector<float> wp((size_t)a.Npad * a.K, 0.f), bp(a.Npad, 0.f);
    plan::pack_conv(Wa->d.data(), k, cin, cout_a, tin.Cs, false, h->bf16, a.K, 0, wp);
    plan::pack_conv(Wb->d.data(), k, cin, cout_b, tin.Cs, false, h->bf16, a.K, cout_a, wp);
    for (int n = 0; n < cout_a; n++) bp[n] = Ba->d[n];
    for (int n = 0; n < cout_b; n++) bp[cout_a + n] = Bb->d[n];
    const int head = (k == 1 && !relu_b && !h->x3 && !getenv("VNECT_NO_HEAD_SPLIT") && !getenv("VNECT_FORCE_TILE"))
                         ? plan::pair_head_cols(a.M, cout_a, cout_b, a.K, h->bf16, conv_cu_count()) : 0;
    if (head) {
        Layer Hd;  // channels [0, head) of layer a: an ordinary 1x1 launch into the same tensor
        Hd.op = OP_CONV, Hd.name = sa + "[:" + std::to_string(head) + "]", Hd.in = in, Hd.out = L.out;
        ConvArgs& q = Hd.a;
        q = a;
        q.ldc2 = 0, q.split_n = 0, q.relu_cols = head, q.Nvalid = head;
        Hd.dy[0] = Hd.dx[0] = 0;
        Hd.Nreal = head, Hd.Kreal = L.Kreal, Hd.flops = 2.0 * a.M * (double)L.Kreal * head;
        choose_tile(Hd, (long long)a.M);
        q.Npad = round_up(head, Hd.BN);
        std::vector<float> wh(wp.begin(), wp.begin() + (size_t)head * a.K), bh(bp.begin(), bp.begin() + head);
        wh.resize((size_t)q.Npad * a.K, 0.f), bh.resize(q.Npad, 0.f);
        if (upload_layer_weights(h, Hd, wh) || upload(h, &Hd.bias, bh)) return -1;
        h->layers.push_back(Hd);
        // ... and the pair keeps the rest
        wp.erase(wp.begin(), wp.begin() + (size_t)head * a.K), bp.erase(bp.begin(), bp.begin() + head);
        L.out_col0 = head, L.Nreal -= head, a.Nvalid = L.Nreal, a.Npad -= head;
        a.split_n = cout_a - head, a.relu_cols = cout_a - head;
        L.flops = 2.0 * a.M * (double)L.Kreal * L.Nreal;
    }
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp)) return -1;
    if (k == 1 && stride == 1 && cin == 64 && tin.Cs == 64 && cout_a == 64 && cout_b == 256) {
        // res2a_branch2a + res2a_branch1 read pool1: the stem can run them on its pooled tile (setup_stem) and wants the weights in
        // fragment order, both layers side by side ([k][n] with n over the 64 + 256 outputs)
        std::vector<float> cat((size_t)64 * 320), fw;
        for (int kk = 0; kk < 64; kk++) {
            for (int n = 0; n < 64; n++) cat[(size_t)kk * 320 + n] = Wa->d[(size_t)kk * 64 + n];
            for (int n = 0; n < 256; n++) cat[(size_t)kk * 320 + 64 + n] = Wb->d[(size_t)kk * 256 + n];
        }
        plan::pack_tail(cat.data(), 64, 320, h->bf16, fw);
        if (upload_weights(h, &L.frag_w, fw)) return -1;
    }
    h->layers.push_back(L);
    *second = L.out2;
    return L.out;
}

// A 3x3 conv whose 64 output channels feed a 1x1 conv (+ shortcut + ReLU) of the same pixels -- res2*_branch2b -> res2*_branch2c,
// vnect_model.py:38-41,50-53,56-59 -- as ONE launch: the 3x3 layer's tile stays in LDS and the 1x1 layer is a second GEMM
// inside the workgroup (conv.hip, TAIL).  Possible where a workgroup owns ALL of the 3x3 layer's channels for its rows (N = 64:
// the 92x92 stage) and has one tile (so the ring is free behind the K loop): ceil(M / 64) <= 512 workgroups.  Results are
// bit-identical to the two stand-alone launches.  Returns the block output tensor or -1; *fits = false if the shape does not
// admit the fusion (the caller then builds the two layers).
//
// The WIDE form (round 3) does the same for a 3x3 layer with 128 channels -- res3*_branch2b -> res3*_branch2c and the head's
// res5c_branch2b -> res5c_branch2c (vnect_model.py:62-103,211-217): the 3x3 layer runs on 32 x 128 tiles (as many workgroups as the
// 64 x 64 plan has tiles, the same MFMA work per wave), a workgroup owns all 128 channels of its 32 pixels, the 1x1 layer (K = 128,
// up to 512 outputs) is conv.hip's tail_wide.  One workgroup per CU (100-KB ring): ceil(M / 32) <= 256, i.e. up to three scales at
// 46x46.  Not on a split-product handle (its 3x3 layers keep the 64 x 64 split-product loop).  VNECT_NO_WIDE_TAIL=1: A/B runs.
static int add_conv_tail(vnect_handle* h, const std::string& sb, const std::string& sc, int in, int resid, const std::string& out_name,
                  int mid, int cout, bool* fits, bool relu2 = true, const std::string& chain_scope = "", int* chain_out = nullptr)
{
    const Tensor tin = h->tensors[in];
    const int EPR = h->bf16 ? 64 : 32;
    const long long pixels = (long long)tin.S * tin.H * tin.W;
    const int cus = conv_cu_count();  // 256 on MI355X: the literals of the comments above
    const bool narrow = mid == 64 && cout == 256 && (pixels + 63) / 64 <= 2 * cus;
    // ... and at least half a chip of them in fp32: one scale at 46x46 is 67 workgroups, each alone with the whole 3x3 K loop where the
    // stand-alone layer splits K over the idle CUs -- measured, single-scale handle (a pyramid rank): 0.670 ms per frame with the wide
    // tails against 0.619 without; two scales (133): level; bf16: level at one scale, +5 % at two (tools/one_scale_ab.py).
    const long long wide_wgs = (pixels + 31) / 32;
    const bool wide = mid == 128 && cout <= 512 && wide_wgs <= cus && (h->bf16 || wide_wgs >= cus / 2 || getenv("VNECT_FORCE_WIDE_TAIL")) && !h->x3 &&
                      !getenv("VNECT_NO_WIDE_TAIL") && !getenv("VNECT_FORCE_TILE");
    *fits = (narrow || wide) && tin.Cs % EPR == 0 && !h->keep_activations && !getenv("VNECT_NO_TAIL");
    if (!*fits) return -1;
    const int cin = tin.C;
    const HostArray* Wb = get_w(h, sb + "/weights", {3, 3, cin, mid});
    const HostArray* Bb = Wb ? get_w(h, sb + "/biases", {mid}) : nullptr;
    const HostArray* Wc = Bb ? get_w(h, sc + "/weights", {1, 1, mid, cout}) : nullptr;
    const HostArray* Bc = Wc ? get_w(h, sc + "/biases", {cout}) : nullptr;
    if (!Bc) return -1;
    int ho, wo, pt = 0, pl = 0;
    same_pad(tin.H, 3, 1, &ho, &pt), same_pad(tin.W, 3, 1, &wo, &pl);
    Layer L;
    L.op = OP_CONV, L.name = sb + ">" + sc, L.in = in, L.resid = resid;
    const bool final_maps = sc == "res5c_branch2c";  // feeds the f64 post-processing: stays fp32
    L.out = add_tensor(h, out_name, tin.S, ho, wo, cout, cout, final_maps);
    ConvArgs& a = L.a;
    a.out_f32 = final_maps;
    a.S = tin.S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
    a.Ho = ho, a.Wo = wo, a.M = tin.S * ho * wo, a.stride = 1;
    a.OH = ho, a.OW = wo, a.os = 1, a.nphase = 1;
    a.ldc = cout, a.ldr = cout, a.relu_cols = relu2 ? cout : 0, a.Nvalid = cout;  // the TAIL's output, shortcut and ReLU
    a.bf16 = h->bf16;
    a.ntaps = 9, a.cpt = tin.Cs / EPR, a.K = 9 * tin.Cs;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) L.dy[ky * 3 + kx] = ky - pt, L.dx[ky * 3 + kx] = kx - pl;
    L.Nreal = mid, L.Kreal = 9 * cin;
    L.flops = 2.0 * a.M * ((double)L.Kreal * mid + (double)mid * cout);
    L.BM = wide ? 32 : 64, L.BN = mid, L.KG = 1, a.ksplit = 1;
    a.Npad = mid;
    std::vector<float> wp((size_t)mid * a.K, 0.f), bp(mid, 0.f), w2, b2(round_up(cout, 32), 0.f);
    plan::pack_conv(Wb->d.data(), 3, cin, mid, tin.Cs, false, h->bf16, a.K, 0, wp);
    for (int n = 0; n < mid; n++) bp[n] = Bb->d[n];
    plan::pack_tail(Wc->d.data(), mid, cout, h->bf16, w2);
    for (int n = 0; n < cout; n++) b2[n] = Bc->d[n];
    float *dw2 = nullptr, *db2 = nullptr;
    if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp) || upload_weights(h, &dw2, w2) || upload(h, &db2, b2)) return -1;
    a.tail_w = dw2, a.tail_bias = db2, a.tail_n = cout;
    // Chain GEMM (conv.hip: chain_gemm): the NEXT block's branch2a (1x1, 512 -> 128, ReLU) on the output tile while it is still in LDS --
    // one launch fewer per identity block of the 46x46 stage.  bf16 only: there a launch is mostly fixed cost (+2.9 % frames/s, A/B in
    // one call); in fp32 the chained layer is MFMA-bound either way (7.6 us of matrix work on the same four SIMDs) and the chain's own
    // overhead exceeds the launch it saves (-1.5 %).  VNECT_NO_CHAIN=1 / VNECT_FORCE_CHAIN=1: A/B runs and the fp32 form's parity test.
    if (chain_out) *chain_out = -1;
    // The 64-wide tail chains too (res2a -> res2b_branch2a, 256 -> 64; conv.hip: chain_narrow), in bf16 only: its output tile would not
    // fit the two-workgroups-per-CU ring in fp32.
    const bool chain_wide = wide && cout == 512 && (h->bf16 || getenv("VNECT_FORCE_CHAIN"));
    const bool chain_narrow = narrow && !wide && cout == 256 && h->bf16 && resid >= 0;
    if ((chain_wide || chain_narrow) && relu2 && !chain_scope.empty() && chain_out && !getenv("VNECT_NO_CHAIN")) {
        const int cn = chain_wide ? 128 : 64;
        const HostArray* Wn = get_w(h, chain_scope + "/weights", {1, 1, cout, cn});
        const HostArray* Bn = Wn ? get_w(h, chain_scope + "/biases", {cn}) : nullptr;
        if (!Bn) return -1;
        std::vector<float> w3, b3(Bn->d.begin(), Bn->d.end());
        plan::pack_tail(Wn->d.data(), cout, cn, h->bf16, w3);
        float *dw3 = nullptr, *db3 = nullptr;
        if (upload_weights(h, &dw3, w3) || upload(h, &db3, b3)) return -1;
        L.out3 = add_tensor(h, chain_scope, tin.S, ho, wo, cn, cn);
        a.chain_w = dw3, a.chain_bias = db3, a.chain_n = cn, a.chain_ld = cn;
        L.name += ">" + chain_scope;
        L.flops += 2.0 * a.M * (double)cout * cn;
        *chain_out = L.out3;
    }
    h->layers.push_back(L);
    return L.out;
}

// point a conv layer's arguments at h's activation buffers and workspace (weights are whatever L already holds)
void bind_activations(vnect_handle* h, Layer& L)
{
    ConvArgs& a = L.a;
    a.in = h->tensors[L.in].d;
    a.out = (float*)((char*)h->tensors[L.out].d + (size_t)L.out_col0 * h->tensors[L.out].esz);
    a.out2 = L.out2 >= 0 ? h->tensors[L.out2].d : nullptr;
    a.chain_out = L.out3 >= 0 ? h->tensors[L.out3].d : nullptr;
    a.resid = L.resid >= 0 ? h->tensors[L.resid].d : nullptr;
    a.w = L.w, a.bias = L.bias, a.scale = L.scale, a.shift = L.shift, a.ws = h->ws;
    L.r.ws = h->ws, L.r.resid = a.resid, L.r.out = a.out;
}

// The fused stem (stem.hip) stands for layers l_conv1 + l_pool1 (and, from the frame, for pyramid_kernel).  Default: from the frame
// on handles whose layers share the arena; handles with per-layer read-back keep the stand-alone layers (their "conv1" activation
// must exist) unless VNECT_FORCE_STEM says otherwise (the parity test reads pool1 from both forms).  VNECT_NO_STEM=1 restores the
// three launches, VNECT_STEM=batch keeps pyramid_kernel and fuses conv1 + pool1 only (A/B runs).
void setup_stem(vnect_handle* h)
{
    h->stem_mode = 0, h->stem_pair = false;
    if (h->l_conv1 < 0 || h->l_pool1 != h->l_conv1 + 1) return;
    const Layer& C = h->layers[h->l_conv1];
    const Tensor& tin = h->tensors[h->t_input4];
    const Tensor& tp = h->tensors[h->layers[h->l_pool1].out];
    if (tin.H != BOX || tin.W != BOX || tp.H != 92 || tp.W != 92 || tp.Cs != 64 || C.a.Npad != 64 || C.a.ksplit != 1) return;
    const char* force = getenv("VNECT_FORCE_STEM");
    const char* mode = getenv("VNECT_STEM");
    if (getenv("VNECT_NO_STEM")) return;
    if (h->keep_activations && !force) return;
    if (force) mode = force;
    h->stem_mode = (mode && !strcmp(mode, "batch")) ? 1 : 2;
    StemArgs& a = h->stem;
    memset(&a, 0, sizeof a);
    a.batch = tin.d, a.w = C.w, a.bias = C.bias, a.out = tp.d;
    a.fp = h->d_fp, a.tabs = h->d_stabs;
    a.S = h->Snet, a.scale_base = h->sharded ? h->cfg.pyramid_rank : 0, a.bf16 = h->bf16;
    // row groups of 4 and 5 pooled rows (hostplan.h)
    a.groups = plan::stem_groups(a.S, a.row0);
    // (a lane's tables are lane 0's: build_twin copies stabs_host before calling this)
    h->stem_frame_ok = h->stem_mode == 2 && plan::stem_frame_fits(h->stabs_host, a.S, a.scale_base, a.groups, a.row0, h->bf16);
    // PAIR form: the launch behind pool1 is res2a_branch2a + res2a_branch1 (1x1 on pool1's 64 channels) and nothing else reads pool1:
    // the stem runs it on the pooled tile and pool1 is never written.  Not with per-layer read-back (pool1 must exist there).
    // VNECT_NO_STEM_PAIR=1: A/B runs.
    h->stem_pair = false;
    const size_t lp = (size_t)h->l_pool1 + 1;
    if (!h->keep_activations && !getenv("VNECT_NO_STEM_PAIR") && lp < h->layers.size()) {
        const Layer& P = h->layers[lp];
        if (P.op == OP_CONV && P.frag_w && P.in == h->layers[h->l_pool1].out && P.out >= 0 && P.out2 >= 0 && P.a.Npad == 320 &&
            P.a.split_n == 64 && P.a.relu_cols == 64 && P.a.ldc == 64 && P.a.ldc2 == 256 && P.a.M == a.S * 92 * 92) {
            a.pair_w = P.frag_w, a.pair_bias = P.bias, a.pair_out_a = h->tensors[P.out].d, a.pair_out_b = h->tensors[P.out2].d;
            h->stem_pair = true;
        }
    }
}

int finalize_impl(vnect_handle* h)
{
    const int S = h->Snet;
    h->tensors.clear(), h->layers.clear(), h->tensor_by_name.clear();
    h->t_input4 = add_tensor(h, "input", S, BOX, BOX, 3, 4);
    auto conv = [&](const std::string& scope, int in, int k, int stride, int cout, bool relu, int resid = -1,
                    const std::string& out_name = "") {
        ConvSpec sp;
        sp.scope = scope, sp.out_name = out_name.empty() ? scope : out_name;
        sp.in = in, sp.resid = resid, sp.k = k, sp.stride = stride, sp.cout = cout, sp.relu = relu;
        return add_conv(h, sp);
    };
#define NEED(x)                                 \
    do {                                        \
        if ((x) < 0) return VNECT_E_ARG;        \
    } while (0)
    // vnect_model.py:27-29
    int conv1 = conv("conv1", h->t_input4, 7, 2, 64, true);
    NEED(conv1);
    int pool1;
    {
        const Tensor t = h->tensors[conv1];
        int ho, wo, p;
        same_pad(t.H, 3, 2, &ho, &p), same_pad(t.W, 3, 2, &wo, &p);
        Layer L;
        L.op = OP_POOL, L.name = "pool1", L.in = conv1;
        pool1 = L.out = add_tensor(h, "pool1", S, ho, wo, 64, 64);
        h->l_conv1 = (int)h->layers.size() - 1;
        h->layers.push_back(L);
        h->l_pool1 = (int)h->layers.size() - 1;
    }
    // bottleneck blocks (vnect_model.py:31-165); block output tensors are named resNx
    // branch2b (3x3) -> branch2c (1x1, + shortcut s, ReLU): one launch where the tail GEMM fits (add_conv_tail), else two
    // (`next`: the identity block behind this one -- where the fused launch takes the wide form, that block's branch2a rides along
    // as its chain GEMM and `chained` holds its output tensor for ident() to pick up)
    int chained = -1;
    std::string chained_for;
    auto b_then_c = [&](const std::string& p, int a, int mid, int out, int s, const std::string& next = "") {
        if (a < 0) return -1;
        bool fits = false;
        int co = -1;
        const int o = add_conv_tail(h, p + "_branch2b", p + "_branch2c", a, s, p, mid, out, &fits, true, next.empty() ? "" : next + "_branch2a", &co);
        if (fits && co >= 0) chained = co, chained_for = next;
        if (fits) return o;
        const int b = conv(p + "_branch2b", a, 3, 1, mid, true);
        return b < 0 ? -1 : conv(p + "_branch2c", b, 1, 1, out, true, s, p);
    };
    auto proj = [&](const std::string& p, int x, int mid, int out, int stride, const std::string& next = "") {
        int s = -1;
        int a = add_conv_pair(h, p + "_branch2a", mid, p + "_branch1", out, x, stride, &s);
        return b_then_c(p, a, mid, out, s, next);
    };
    auto ident = [&](const std::string& p, int x, int mid, int out, const std::string& next = "") {
        int a = chained_for == p ? chained : conv(p + "_branch2a", x, 1, 1, mid, true);
        chained_for.clear();
        return b_then_c(p, a, mid, out, x, next);
    };
    // 92x92 stage: each block's 3x3 layer has 64 channels, i.e. one 64-wide tile column, so its 1x1 successor can run as a tail
    // GEMM of the same workgroups (add_conv_tail): 3 launches and 3 x 13 MB of intermediate traffic fewer.  Where the shape does
    // not admit it (more than 512 tiles: four or more scales; per-layer read-back requested) the stand-alone layers run, and in
    // the reference's wiring res2b_branch2b / res2c_branch2b -- both read res2b_branch2a -- share one dual-output launch.
    int r;
    int res2_chained = -1;  // the next block's branch2a where the tail launch of this one has produced it (chain GEMM, bf16)
    {
        int s = -1;
        int a = add_conv_pair(h, "res2a_branch2a", 64, "res2a_branch1", 256, pool1, 1, &s);
        NEED(a);
        bool fits = false;
        r = add_conv_tail(h, "res2a_branch2b", "res2a_branch2c", a, s, "res2a", 64, 256, &fits, true, "res2b_branch2a", &res2_chained);
        if (!fits) {
            int b = conv("res2a_branch2b", a, 3, 1, 64, true);
            NEED(b);
            r = conv("res2a_branch2c", b, 1, 1, 256, true, s, "res2a");
        }
        NEED(r);
    }
    if (h->cfg.paper_res2c) {
        for (const char* p : {"res2b", "res2c"}) {
            const std::string P = p;
            int a = res2_chained >= 0 ? res2_chained : conv(P + "_branch2a", r, 1, 1, 64, true);
            res2_chained = -1;
            NEED(a);
            bool fits = false;
            int o = add_conv_tail(h, P + "_branch2b", P + "_branch2c", a, r, P, 64, 256, &fits, true, P == "res2b" ? "res2c_branch2a" : "", &res2_chained);
            if (!fits) {
                int b = conv(P + "_branch2b", a, 3, 1, 64, true);
                NEED(b);
                o = conv(P + "_branch2c", b, 1, 1, 256, true, r, P);
            }
            NEED(o);
            r = o;
        }
    } else {
        // vnect_model.py:50-57: res2c_branch2b consumes res2b_branch2a (`:56`), res2c_branch2a is dead and pruned
        if (!get_w(h, "res2c_branch2a/weights", {1, 1, 256, 64})) return VNECT_E_ARG;  // schema completeness, like the reference's load_weights
        const int x = r;
        int a = res2_chained >= 0 ? res2_chained : conv("res2b_branch2a", x, 1, 1, 64, true);
        res2_chained = -1;
        NEED(a);
        bool fits = false;
        int r2b = add_conv_tail(h, "res2b_branch2b", "res2b_branch2c", a, x, "res2b", 64, 256, &fits);
        if (fits) {
            NEED(r2b);
            r = add_conv_tail(h, "res2c_branch2b", "res2c_branch2c", a, r2b, "res2c", 64, 256, &fits);
            NEED(r);
        } else {
            int b2 = -1;
            int b1 = add_conv_pair(h, "res2b_branch2b", 64, "res2c_branch2b", 64, a, 1, &b2, 3, true);
            NEED(b1);
            r2b = conv("res2b_branch2c", b1, 1, 1, 256, true, x, "res2b");
            NEED(r2b);
            r = conv("res2c_branch2c", b2, 1, 1, 256, true, r2b, "res2c");
            NEED(r);
        }
    }
    r = proj("res3a", r, 128, 512, 2, "res3b");
    NEED(r);
    {
        const char* blocks[] = {"res3b", "res3c", "res3d", ""};
        for (int i = 0; i < 3; i++) {
            r = ident(blocks[i], r, 128, 512, blocks[i + 1]);
            NEED(r);
        }
    }
    r = proj("res4a", r, 256, 1024, 2);
    NEED(r);
    for (const char* p : {"res4b", "res4c", "res4d", "res4e", "res4f"}) {
        r = ident(p, r, 256, 1024);
        NEED(r);
    }
    // res5a / res5b (vnect_model.py:167-185)
    {
        int s = -1;
        int a = add_conv_pair(h, "res5a_branch2a_new", 512, "res5a_branch1_new", 1024, r, 1, &s);
        NEED(a);
        int b = conv("res5a_branch2b_new", a, 3, 1, 512, true);
        NEED(b);
        r = conv("res5a_branch2c_new", b, 1, 1, 1024, true, s, "res5a");
        NEED(r);
        a = conv("res5b_branch2a_new", r, 1, 1, 256, true);
        NEED(a);
        // (round 6, measured and dropped: in bf16 these two as ONE wide-tail launch on 32 x 128 tiles -- 50 workgroups owning all 128 mid
        // channels of their rows -- take 10.9 us against 5.7 + 2.1 us and a boundary: NEGATIVE_RESULTS.md)
        b = conv("res5b_branch2b_new", a, 3, 1, 128, true);
        NEED(b);
        r = conv("res5b_branch2c_new", b, 1, 1, 256, true);
        NEED(r);
    }
    // Transposed convs + BN + ReLU + deltas, one 4-phase launch (vnect_model.py:188-209).
    // out[2i-1+ky, 2j-1+kx, oc] += in[i,j,ic] * W[ky,kx,oc,ic]; phase (py,px) = (oy&1, ox&1):
    //   py = 0: ky = 1 reads row i', ky = 3 reads row i'-1;  py = 1: ky = 0 reads row i'+1, ky = 2 reads row i'.
    int feat;
    {
        const Tensor tin = h->tensors[r];
        const HostArray* W1 = get_w(h, "res5c_branch1a/kernel", {4, 4, 63, 256});
        const HostArray* W2 = get_w(h, "res5c_branch2a/kernel", {4, 4, 128, 256});
        const HostArray* ga = get_w(h, "bn5c_branch2a/gamma", {128});
        const HostArray* be = get_w(h, "bn5c_branch2a/beta", {128});
        const HostArray* mu = get_w(h, "bn5c_branch2a/moving_mean", {128});
        const HostArray* va = get_w(h, "bn5c_branch2a/moving_variance", {128});
        if (!W1 || !W2 || !ga || !be || !mu || !va) return VNECT_E_ARG;
        Layer L;
        L.op = OP_CONV, L.name = "res5c_deconv", L.in = r;
        const int featCs = h->bf16 ? 256 : 224;  // 212 channels padded to a whole number of K chunks
        feat = L.out = add_tensor(h, "res5c_branch2a_feat", S, 2 * tin.H, 2 * tin.W, 212, featCs);
        ConvArgs& a = L.a;
        a.S = S, a.H = tin.H, a.W = tin.W, a.Cs = tin.Cs;
        a.Ho = tin.H, a.Wo = tin.W, a.M = S * tin.H * tin.W;
        a.stride = 1, a.OH = 2 * tin.H, a.OW = 2 * tin.W, a.os = 2, a.nphase = 4;
        a.ntaps = 4, a.cpt = tin.Cs / (h->bf16 ? 64 : 32), a.K = 4 * tin.Cs;
        a.bf16 = h->bf16;
        a.ldc = featCs, a.ldr = 0, a.relu_cols = 128, a.Nvalid = 191;
        L.Nreal = 191, L.Kreal = 4 * 256;
        L.flops = 2.0 * (double)S * 46 * 46 * 4 * 256 * 191;
        choose_tile(L, (long long)S * 46 * 46, !h->x3);
        a.Npad = round_up(191, L.BN);
        a.w_phase_stride = (long long)a.Npad * a.K;
        std::vector<float> wp, bp, sc, sh;
        plan::pack_deconv(W1->d.data(), W2->d.data(), a.Npad, a.K, wp, L.dy, L.dx);
        // FusedBatchNorm inference (contrib batch_norm default epsilon 0.001): (x - mean) * (gamma * rsqrt(var + eps)) + beta
        plan::fold_bn(ga->d.data(), be->d.data(), mu->d.data(), va->d.data(), 128, a.Npad, bp, sc, sh);
        if (upload_layer_weights(h, L, wp) || upload(h, &L.bias, bp) || upload(h, &L.scale, sc) || upload(h, &L.shift, sh))
            return VNECT_E_HIP;
        // bone-length features (vnect_model.py:198-209): inside this launch (conv.hip, FUSE = 2) where every workgroup has one tile,
        // i.e. up to 5 scales; as a launch of their own otherwise, and when per-layer read-back is requested
        const bool shape96 = L.BM == 64 && L.BN == 96 && L.KG == 2;
        const long long deconv_items = (long long)((a.M + L.BM - 1) / L.BM) * (a.Npad / L.BN) * 4;
        const bool fuse_bone = ((L.BM == 64 && L.BN == 64 && L.KG == 1 && deconv_items <= 2 * conv_cu_count()) || (shape96 && deconv_items <= conv_cu_count())) && a.ksplit == 1 &&
                               !h->keep_activations && !getenv("VNECT_NO_BONE_FUSE");
        a.bone = fuse_bone;
        if (fuse_bone) L.name = "res5c_deconv+bone_length";
        h->layers.push_back(L);
        if (!fuse_bone) {
            Layer Bn;
            Bn.op = OP_BONE, Bn.name = "res5c_bone_length", Bn.in = feat, Bn.out = feat;
            h->layers.push_back(Bn);
        }
    }
    // head (vnect_model.py:211-217)
    {
        const HostArray* Wk = get_w(h, "res5c_branch2c/kernel", {1, 1, 128, 84});
        if (!Wk) return VNECT_E_ARG;
        // tf.layers.conv2d without bias == the tc.layers form with zero biases
        HostArray z;
        z.d.assign(84, 0.f), z.shape = {84};
        h->weights["res5c_branch2c/weights"] = *Wk;
        h->weights["res5c_branch2c/biases"] = z;
        bool fits = false;
        h->t_out = add_conv_tail(h, "res5c_branch2b", "res5c_branch2c", feat, -1, "res5c_branch2c", 128, 84, &fits, false);
        if (!fits) {
            const int hd = conv("res5c_branch2b", feat, 3, 1, 128, true);
            h->t_out = hd < 0 ? -1 : conv("res5c_branch2c", hd, 1, 1, 84, false);
        }
        h->weights.erase("res5c_branch2c/weights"), h->weights.erase("res5c_branch2c/biases");
        NEED(h->t_out);
    }
#undef NEED
    // buffers
    // + 64 pixels of slack per tensor: the streaming conv kernel's epilogue reads shortcut rows and writes output rows of
    // its last 64-row tile without a per-row bound check (rows >= M land in the slack and are never read)
    auto padded = [](const Tensor& t) { return (t.bytes() + (size_t)64 * t.Cs * t.esz + 255) & ~(size_t)255; };
    if (!h->keep_activations) {
        // Activation arena: a tensor lives from the layer that writes it to the last layer that reads it, and tensors with
        // disjoint lifetimes share addresses (first fit over the live intervals).  The per-frame working set is then the
        // peak live set (~0.1 GB at S = 3) instead of one buffer per layer output (~0.35 GB), so weights + activations stay
        // inside the 256 MiB Infinity Cache from frame to frame.
        const int nt = (int)h->tensors.size(), nl = (int)h->layers.size();
        std::vector<int> first(nt, nl + 1), last(nt, -2);
        auto touch = [&](int t, int l) {
            if (t < 0) return;
            first[t] = std::min(first[t], l), last[t] = std::max(last[t], l);
        };
        touch(h->t_input4, -1);  // written by the pre-processing
        for (int l = 0; l < nl; l++) {
            const Layer& L = h->layers[l];
            touch(L.in, l), touch(L.resid, l), touch(L.out, l), touch(L.out2, l), touch(L.out3, l);
            // the stem may run this pair itself (setup_stem, PAIR form) and then writes its outputs while it still reads the batch tensor:
            // they must not share addresses with anything alive from conv1 on
            if (L.frag_w && l == h->l_pool1 + 1) touch(L.out, h->l_conv1), touch(L.out2, h->l_conv1);
        }
        touch(h->t_out, nl);  // read by the post-processing
        std::vector<size_t> need(nt), off;
        for (int t = 0; t < nt; t++) {
            if (last[t] < first[t]) first[t] = -1, last[t] = nl;  // never touched by a layer: keep it private
            need[t] = padded(h->tensors[t]);
        }
        const size_t total = plan::arena_first_fit(first, last, need, off);
        char* base = nullptr;
        int rc = dev_alloc(h, &base, total);
        if (rc) return rc;
        HIPCK(h, hipMemset(base, 0, total));
        for (int t = 0; t < nt; t++) h->tensors[t].d = (float*)(base + off[t]);
        h->arena_bytes = total, h->arena_off = off;
    } else {
        for (Tensor& t : h->tensors) {
            char* p = nullptr;
            int rc = dev_alloc(h, &p, padded(t));
            if (rc) return rc;
            t.d = (float*)p;
            HIPCK(h, hipMemset(t.d, 0, padded(t)));
        }
    }
    size_t ws = 0;
    for (Layer& L : h->layers)
        if (L.op == OP_CONV && L.a.ksplit > 1)
            ws = std::max(ws, (size_t)L.a.ksplit * ((size_t)L.a.S * L.a.OH * L.a.OW + 64) * L.a.Npad);
    h->ws_floats = ws;
    if (ws) {
        int rc = dev_alloc(h, &h->ws, ws);
        if (rc) return rc;
    }
    HIPCK(h, hipDeviceSynchronize());
    h->conv_flops = 0, h->conv_launches = 0;
    for (Layer& L : h->layers) {
        if (L.op != OP_CONV) continue;
        ConvArgs& a = L.a;
        bind_activations(h, L);
        {   // tap byte offsets for the buffer-addressed loads (kernels.h)
            const int esz = a.bf16 ? 2 : 4, nt = a.nphase * a.ntaps;
            int lo = 0;
            for (int t = 0; t < nt; t++) lo = std::min(lo, (L.dy[t] * a.W + L.dx[t]) * a.Cs * esz);
            a.tap_bias = -lo;
            a.tapgrid = 0;
            if (a.nphase == 1 && a.ntaps == 1 && L.dy[0] == 0 && L.dx[0] == 0 && !a.pixmode) a.tapgrid = 1;
            if (a.nphase == 1 && a.ntaps == 9 && !a.pixmode) {
                bool ok = true;
                for (int t = 0; t < 9; t++) ok = ok && L.dy[t] == t / 3 - 1 && L.dx[t] == t % 3 - 1;
                if (ok) a.tapgrid = 3;
            }
            a.dy_pack = a.dx_pack = 0;
            for (int t = 0; t < nt; t++) {
                if (L.dy[t] < -8 || L.dy[t] > 7 || L.dx[t] < -8 || L.dx[t] > 7) {
                    h->err = "internal: filter tap outside the packed range";
                    return VNECT_E_ARG;
                }
                a.dy_pack |= (unsigned long long)(L.dy[t] + 8) << (4 * t), a.dx_pack |= (unsigned long long)(L.dx[t] + 8) << (4 * t);
            }
        }
        if (a.ksplit > 1) {
            ReduceArgs& q = L.r;
            q.bias = L.bias, q.scale = L.scale, q.shift = L.shift;
            a.slab_pix = (long long)a.S * a.OH * a.OW + 64, q.slab_pix = a.slab_pix;
            q.npix = (long long)a.S * a.OH * a.OW, q.Npad = a.Npad, q.Nvalid = a.Nvalid, q.ldc = a.ldc, q.ldr = a.ldr;
            q.ksplit = a.ksplit, q.relu_cols = a.relu_cols;
            q.bf16 = a.bf16, q.out_f32 = a.out_f32;
        }
        h->conv_flops += L.flops;
        h->conv_launches += 1;
    }
    setup_stem(h);
    if (h->stem_pair) h->conv_launches -= 1;
    return VNECT_OK;
}

}  // namespace rt
}  // namespace vnect
