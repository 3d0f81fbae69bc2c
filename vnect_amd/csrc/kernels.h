// kernels.h -- argument blocks and launchers shared by the HIP kernels and the host runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tables.h"

namespace vnect {

constexpr int MAX_TAPS = 16;
constexpr int ARG_SLABS_MAX = 32;  // upper bound of the arg-max workgroups per joint (post.hip: ARG_SLABS): what the partials' buffer is sized for
constexpr int PROF_WGS = 512;   // end-stamp slots per launch (two workgroups per CU; a power of two: larger grids wrap around)
constexpr int PROF_SLOTS = 28;  // u64 per layer in the profiling buffer: [0] min start, [1..8] max end per id&7, [9..23] tuning stamps
                                // (vnect_get_layer_stamps), [24..26] workgroup 0: shader-clock counter (s_memtime) at its start and at its
                                // end, 100 MHz stamp at its end -- the clock the chip held during the launch (round 6)

// Implicit-GEMM convolution: out[m][n] = sum_k A[m][k] * Wp[n][k],
//   m = (s*Ho + oy)*Wo + ox, k = (tap, ci), A[m][k] = in[s][oy*stride + dy[tap]][ox*stride + dx[tap]][ci]
struct ConvArgs {
    const float* in;
    const float* w;       // packed [phase][Npad][K], K contiguous
    const float* bias;    // [Npad] (zeros when the layer has none)
    const float* scale;   // [Npad] or nullptr: v = (acc + bias) * scale + shift
    const float* shift;
    const float* resid;   // nullptr or same pixel indexing as out
    float* out;
    float* out2;          // second output tensor for columns >= split_n (two layers sharing one input), or nullptr
    float* ws;            // split-K workspace [ksplit][slab_pix][Npad]
    long long slab_pix;   // pixels per slab: out pixels + 64 of slack (the last tile's rows past M land there)
    unsigned long long* prof;  // nullptr, or {min start, ...} of this launch in 100 MHz s_memrealtime ticks
    unsigned long long* prof_end;  // profiling twin: [PROF_WGS] end stamp of every workgroup (plain stores: an atomic max
                                   // over 500 workgroups put ~1.4 us behind every launch); the host takes the maximum
    int S, H, W, Cs;      // input grid, floats per input pixel
    int Ho, Wo, M;        // logical output grid, M = S*Ho*Wo
    int K, ntaps, cpt;    // K = ntaps*cpt*32
    int stride;
    int Npad, Nvalid;
    int ldc, ldr;         // floats per out / resid pixel
    int ldc2, split_n;    // out2 pixel stride; first column of out2 (multiple of the tile width)
    int OH, OW, os;       // out pixel = (s, oy*os + py, ox*os + px) in an (OH, OW) grid
    int nphase;           // 1, or 4 for the transposed conv (py = phase>>1, px = phase&1)
    int ksplit;
    int relu_cols;        // ReLU on columns < relu_cols
    int pixmode;          // conv1: a chunk row is 8 consecutive NHWC4 pixels of one input row (bf16: of two rows)
    int bf16;             // operands and activations are bf16 (accumulators, bias, slabs stay fp32)
    int out_f32;          // bf16 path: this layer writes fp32 (the final maps feed the f64 post-processing)
    int tiles_m, tiles_n; // filled by the launcher
    int items;            // work items = tiles_m * tiles_n * nphase * ksplit (launcher; streaming kernel)
    unsigned mg_ks, mg_cpt;  // same for d = ksplit, cpt
    unsigned mg_wo, mg_ho, mg_tn, mg_tm;  // ceil(2^32 / d) for d = Wo, Ho, tiles_n, tiles_m (launcher): x / d == umulhi(x, mg) for x*d < 2^32
    long long w_phase_stride;
    // streaming kernel (buffer-addressed LDS-DMA): the byte offset of a tap from the lane's input pixel is
    // (dy*W + dx)*Cs*esz + tap_bias with tap_bias = -min over taps (so it is >= 0; the descriptor's base is in - tap_bias)
    int tap_bias;
    // filter taps (dy, dx) as 4-bit fields (value + 8, entry [phase*ntaps + tap] at bits 4*entry): the producers decode them
    // with scalar shifts -- no table in the argument block, no dependent scalar load per tap
    unsigned long long dy_pack, dx_pack;
    int tapgrid;  // 1: single tap (0,0); 3: the 3x3 grid with pad 1 (validity masks in closed form); 0: walk the table
    // Tail GEMM (conv_stream_kernel<..., FUSE = 1>; ONE tile per workgroup that holds ALL of the layer's channels: 64x64 tiles for
    // N = 64, 32x128 tiles for N = 128): the layer's own output never reaches HBM -- relu(acc + bias) stays in LDS as a tile and feeds
    // a 1x1 conv `tail_w` (K = the layer's N; in MFMA fragment order, hostplan.h: pack_tail) whose bias / shortcut / ReLU / output are
    // the fields `tail_bias`, resid, relu_cols, out, ldc, Nvalid above describe.  (res2*_branch2b -> res2*_branch2c, res3*_branch2b ->
    // res3*_branch2c, res5c_branch2b -> res5c_branch2c: vnect_model.py:38-41,50-53,56-59,62-103,211-217.)
    const float* tail_w;
    const float* tail_bias;
    int x3;       // 1: split-product form (VNECT_FP32_SPLIT): `w` holds [Npad][K / 32][3 planes][32] bf16 (hostplan.h: pack_split3), the kernel
                  // splits the fp32 activations three ways in registers and multiplies piece by piece on the bf16 matrix pipe
    int bone;     // 1: FUSE = 2 launch of the transposed conv: the bone-length columns 191 .. 211 (+ zero padding) are written here too
    int tail_n;   // 0: no tail; else the 1x1 conv's output channels (64-wide tile: 256 = 2 row halves x 8 column blocks over the 8 waves; 128-wide: up to 512, tail wave tw takes blocks tw, tw + 8)
    // Chain GEMM behind the wide tail (conv.hip: chain_gemm): the tail's output tile -- a bottleneck block's output, 32 pixels x 512
    // channels -- stays in LDS as well and feeds the NEXT block's branch2a (1x1, 512 -> 128, ReLU): `chain_w` (fragment order),
    // `chain_bias`, output `chain_out` ([M][chain_ld]).  res3*_branch2c -> res3(*+1)_branch2a, vnect_model.py:72-103.
    const float* chain_w;
    const float* chain_bias;
    float* chain_out;
    int chain_n;  // 0: none; else 128
    int chain_ld;
};

struct ReduceArgs {  // split-K second pass: out = epilogue(sum_ks ws[ks])
    const float* ws;
    long long slab_pix;  // pixels per slab (ConvArgs::slab_pix)
    const float* bias;
    const float* scale;
    const float* shift;
    const float* resid;
    float* out;
    long long npix;
    int Npad, Nvalid, ldc, ldr, ksplit, relu_cols;
    int bf16, out_f32;
};

// Write-through (`sc1`) vector stores for the small kernels between the conv launches: like the conv epilogue's, their output
// leaves the L2 while the kernel runs instead of in the write-back at the kernel boundary the next launch waits behind.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename V>
__device__ __forceinline__ void store_wt(V* p, V v)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "one dwordx4 / dwordx2 store");
    // (the s_nop: gfx9 requires wait states between a VMEM store of more than 64 bits and a VALU write of its data registers;
    // the compiler inserts them for its own stores but cannot see inside an asm block -- a thread that went on computing
    // right behind such a store stored the next iteration's loop counter instead of its data: found in round 2)
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
#else
template <typename V>
__device__ __forceinline__ void store_wt(V* p, V v) { *p = v; }
#endif

hipError_t launch_conv(const ConvArgs& a, int BM, int BN, int KG, hipStream_t st);  // KG: in-workgroup K groups (1, 2, 4)
hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st);
hipError_t conv_setup();  // one-time function attributes (dynamic LDS size)
bool conv_deconv96_available();  // after conv_setup: the 64x96x2 three-accumulator instantiations fit the register file
int conv_cu_count();             // after conv_setup: compute units of the device (what a "round over the chip" is counted in)
// compile-time probe switches of the kernel translation units, as text and as "all at their product values" (vnect_build_info)
const char* conv_build_probes();
bool conv_probes_off();
const char* post_build_probes();
bool post_probes_off();

hipError_t launch_pad3to4(const float* in3, void* out4, long long npix, int bf16, hipStream_t st);
hipError_t launch_strip4to3(const void* in4, float* out3, long long npix, int bf16, hipStream_t st);
hipError_t launch_maxpool(const void* in, void* out, int S, int H, int W, int C, int Ho, int Wo, int bf16, hipStream_t st);
hipError_t launch_bone(void* feat, long long npix, int ld, int bf16, hipStream_t st);

// ---- pre-processing (the table structs live in tables.h) -------------------------------
struct FrameDyn {     // what does change every frame: passed to the two kernels that need it BY VALUE (kernel arguments),
                      // so a frame costs no host-to-device copy
    double t2d, t3d;
    const uint8_t* frame;  // device pointer
    long long row_stride;
    // pyramid-sharded handle, exchange by peer writes: device word that exchange_kernel sets to the frame's sequence number when a
    // peer's maps did not arrive (nullptr otherwise); post_kernel then skips the joints stage of frame `xseq`
    const unsigned* xfail;
    unsigned xseq;
};

hipError_t launch_pyramid(const FrameParams* fp, FrameDyn dyn, const ScaleTabs* tabs, void* batch4, int S, int scale_base, int bf16, hipStream_t st);

// vnect_infer: H rows of `row` bytes from device-mapped pinned host memory (`stride` bytes apart) into a resident frame slot, as a kernel
hipError_t launch_frame_copy(const uint8_t* src_dev, uint8_t* dst, int H, int row, long long stride, const uint8_t* src_end, hipStream_t st);

// ---- the stem as one launch (stem.hip): [gen_input_batch ->] conv1 + ReLU -> 3x3 / stride-2 max-pool on spatial tiles -------------
struct StemArgs {
    const void* batch;     // (S,368,368,4) NHWC4 batch (fp32 / bf16), or nullptr with from_frame
    const FrameParams* fp; // from_frame: the three arguments of pyramid_kernel
    FrameDyn dyn;
    const ScaleTabs* tabs;
    const float* w;        // conv1's packed weights as the stand-alone layer has them: fp32 [64][7 rows][8 px][4 ch], bf16 [64][4 row pairs][2][8][4]
    const float* bias;     // [64]
    void* out;             // pool1 (S,92,92,64), fp32 / bf16
    // PAIR form (round 3): res2a_branch2a (1x1, 64 -> 64, ReLU) + res2a_branch1 (1x1, 64 -> 256), vnect_model.py:32-35, on the pooled
    // tile while it is in LDS -- pool1 is then never written: weights of both layers side by side in MFMA fragment order
    // (hostplan.h: pack_tail over the concatenated [64][320]), biases [320], the two output tensors (S,92,92,64) / (S,92,92,256)
    const float* pair_w;   // nullptr: no pair
    const float* pair_bias;
    void* pair_out_a;
    void* pair_out_b;
    unsigned long long* prof;      // profiling twin: start stamp (workgroup 0) ...
    unsigned long long* prof_end;  // ... and every workgroup's end stamp, like ConvArgs
    int S, groups;         // images; row groups per image (grid = S * groups * 4 tiles)
    int scale_base;        // from_frame: image 0 of the batch is scale `scale_base` (a pyramid-sharded rank)
    int bf16, from_frame;
    int dbg;               // tuning only (VNECT_STEM_DBG): 1 = no conv blocks, 2 = no pooling, 4 = no patch (timing breakdowns; wrong results)
    unsigned char row0[STEM_MAXGROUPS + 1];  // first pooled row of every group; row0[groups] = 92
};
hipError_t launch_stem(const StemArgs& a, hipStream_t st);
hipError_t stem_setup();  // one-time function attributes (dynamic LDS size)

// ---- post-processing ------------------------------------------------------------------
struct ArgPartial {
    double v;
    int idx;
    int pad_;
};
struct Filt {  // OneEuroFilter + its two LowPassFilters
    double freq, mincutoff, beta, dcutoff;
    double lasttime;
    double x_y, x_s, dx_y, dx_s;
    int has_last, x_init, dx_init, pad_;
};
struct FilterBank {
    Filt f2[NJ][2];
    Filt f3[NJ][3];
};
struct JointsOut {
    double j2d[NJ * 2];
    float j3d[NJ * 3];
    int status;
};

// merge + arg-max + joints in ONE launch: the last of the 168 arg-max workgroups (agent-scope ticket, zero between launches) runs the
// joints stage
hipError_t launch_post(const float* maps, MergeGeo geo, ArgPartial* part, unsigned* ticket,
                       FilterBank* fb, const FrameParams* fp, FrameDyn dyn, int nep50, JointsOut* out, hipStream_t st);
// multi-scale merge of the heat-maps (in LDS, the rows each workgroup needs) + arg-max of the virtual x8 upsample
hipError_t launch_argmax(const float* maps, MergeGeo geo, ArgPartial* part, hipStream_t st);
// out may be (device-mapped) pinned HOST memory: the kernel's 21x2 + 21x3 results then need no device-to-host copy
hipError_t launch_joints(const ArgPartial* part, const float* maps, MergeGeo geo, FilterBank* fb,
                         const FrameParams* fp, FrameDyn dyn, int nep50, JointsOut* out, hipStream_t st);

// ---- pyramid sharding: the exchange by peer writes (SURVEY 8e "plain peer writes ... into the gather buffer") -----------
// Every rank owns one exchange block [2 parities][nranks slots][XCHG_MAPS floats] + flags[2][8] (u32, one 128-B line per
// parity), reachable by every peer (IPC-mapped across processes, hipDeviceEnablePeerAccess inside one).  One launch per frame:
// workgroup (peer p, chunk c) stores chunk c of this rank's maps into slot `rank` of p's block (16-B system-scope
// write-through stores over xGMI), the last chunk-workgroup per peer publishes flags_p[parity][rank] = seq behind a
// system-scope release; then the same workgroup waits (bounded) for flags_local[parity][p] == seq and copies p's slot from the
// local block into the handle's plain gather buffer, which the post-processing kernels read.  Two parities: a rank can run at
// most one frame ahead of a peer (it needs the peer's maps of frame k to finish frame k).
constexpr int XCHG_MAPS = HM * HM * MAPC;   // 177 744 floats = 710 976 B per rank and frame
constexpr int XCHG_CHUNKS = 8;              // workgroups per peer
constexpr size_t XCHG_FLAG_OFF = (size_t)2 * 8 * XCHG_MAPS * sizeof(float);  // byte offset of the flags in an exchange block
constexpr size_t XCHG_BYTES = XCHG_FLAG_OFF + 2 * 128;
struct XchgArgs {
    const float* src;        // this rank's (46,46,84) maps
    char* block[8];          // every rank's exchange block as THIS device addresses it (block[rank] is the local one)
    float* gather;           // local (nranks, 46,46,84): what the merge / arg-max / joints kernels read
    unsigned* tickets;       // local, [8]: chunk-workgroups of a peer that have finished their stores
    int* status;             // device-mapped pinned host word (one per result-ring slot): set to 1 if a peer's flag did not arrive within the bound
    unsigned* dfail;         // device word: set to `seq` in that case (what post_kernel tests, FrameDyn::xfail)
    int rank, nranks, parity;
    unsigned seq;            // frame sequence number (> 0), the flag value
    unsigned spin_limit;     // polls before giving up (each ~1 us)
};
hipError_t launch_exchange(const XchgArgs& a, hipStream_t st);

// VNectEstimator.joint_filter alone: bank `dim` of fb over NJ * dim values (float64 carriers; f32vals: they are float32 scalars)
hipError_t launch_filter(FilterBank* fb, int dim, bool f32vals, int nep50, double t, const double* in, double* out, hipStream_t st);

}  // namespace vnect
