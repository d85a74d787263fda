// Declarations shared by the convolution translation units (aq_conv.hip: the kernels of the reference's ResNet-20 shapes,
// the general fallback kernels and the C entry points; aq_conv_c16.hip: the many-channel 3x3 / 1x1 kernels).
#pragma once
#include "aq_common.h"
#include "aq_bn_shared.h"
#include "../../include/aq_hip.h"

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace aq {

typedef float floatx4 __attribute__((ext_vector_type(4)));

enum ConvMode { MODE_FWD = 0, MODE_FWD_CODES = 1, MODE_DGRAD = 2 };

// Buffer addressing (raw buffer resource: base pointer + size in bytes): a lane's 32-bit byte offset is
// range-checked by the hardware, loads beyond the size return 0 and stores are dropped, so a predicate
// costs one select on the offset instead of a branch around the access, and per-channel / per-image
// strides ride in the scalar offset.  BUF_OOB is an offset no tensor reaches (sizes are < 2 GiB here).
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

struct ConvArgs {
    const float *in;     // x (forward) or dy (dgrad), NCHW
    const float *W;      // [Cout, Cin, 3, 3]
    const uint8_t *idx;  // codes, same shape as W (MODE_FWD_CODES)
    const float *q;      // levels (MODE_FWD_CODES)
    float *out;
    const float *res;    // optional tensor added to the output (the block's shortcut), same shape as out
    int L;
    int N, Cin, Cout;    // conv definition
    int M, KC, K;        // GEMM: output channels, input channels, 9*KC
    int OH, OW;          // output spatial size of this GEMM
    int IH, IW;          // spatial size of the (undilated) input tensor
    int rs;              // step between neighbouring output pixels inside the LDS image
    int dil;             // 2 when the LDS image is the zero-dilated input (stride-2 dgrad), else 1
    // Stride-2 dgrad by output parity (persistent kernel): dx[2a+py][2b+px] only receives the taps whose row /
    // column parity fits, (1, 2, 2, 4 of the nine for the four classes), read from dy[a + {0,1}][b + {0,1}]; a
    // tile holds pixels of ONE class, TH x TW in class-grid coordinates, and walks only that class's share of
    // the k axis: no zero-dilated image, a quarter of the matrix work.  par = 1: on; tyc = row bands per class
    // (tiles_y = 4 * tyc); halo = 1 - par (rows / columns in front of the tile in the LDS image).
    int par, tyc, halo;
    int TW, TH, tiles_x, tiles_y;
    int pitch, rows_in, plane;
    int Kpad;
    int vecA;            // rows of A are 16-byte aligned and K % 4 == 0
    // forward only: the input is a raw tensor whose batch normalisation + ReLU is applied while it is staged
    // (relu(bn(x)) is never written to memory); statistics come as per-slice partials
    int bn_on;
    BnStatsIn bn;
    BnGeom bng;
    // forward, persistent kernel only: per-workgroup (mean, M2) of every output channel over the pixels the
    // workgroup produced, [M][gridDim.x]; the consumer is the batch normalisation that follows (NULL: off)
    float2 *stat_part;
    // dgrad, persistent kernel only: the output is d loss / d relu(bn(bx)); the epilogue also reduces what the
    // batch-norm backward needs, per workgroup and channel: (sum g, sum g * xhat) with g = out where
    // fma(bx, scale, shift) > 0 else 0, xhat = (bx - mean) * rstd, into bwd_part[M][gridDim.x] (NULL: off)
    const float *bx, *bgamma, *bbeta, *bmean, *brstd;
    float2 *bwd_part;
    // general kernel (conv_gemm_chunked_kernel) only: kernel size, taps per channel, rows / columns of padding in
    // front of the (dilated) input, channels per LDS chunk, k extent of a chunk padded to whole k-blocks
    int ks, ks2, padT, CK, Kc;
};


// The same for up to WGRAD_GROUP_MAX layers of one shape in one launch (the weight gradients of a backward pass depend on
// nothing but their own (x, dy), so they can all wait for its end): layer l owns the blocks [l * gx, (l + 1) * gx) of the
// grid's x axis.  One launch of a single layer fills a fraction of the chip and pays the launch floor alone.
constexpr int WGRAD_GROUP_MAX = 8;
struct WgradGroup {
    const float *x[WGRAD_GROUP_MAX], *dy[WGRAD_GROUP_MAX];
    float *part[WGRAD_GROUP_MAX];
    int n, gx;
    int total;  // layers of the whole call (it may take several launches of n <= WGRAD_GROUP_MAX): what the slab count is planned for
};

// Partials of the output-channel statistics a forward launch leaves behind: workgroup w of P covered the tiles
// w, w + P, .. of ntiles, tile_px pixels each (P == 0: this shape has no statistics epilogue).
struct ConvStatsPlan {
    int P, ntiles, tile_px;
    // the persistent launch itself (persist == 0: another kernel runs this shape)
    int persist, MT, NT, c4, jj, pitch_mul, tw_shift;
    unsigned grid_x, grid_y, k4n_rcp;
    size_t lds;
};
constexpr int CONV_STATS_MAX_PARTS = 512;  // the consumer keeps ceil(P / 64) partials per lane in registers


inline int conv_out(int H, int ks, int stride, int pad) { return (H + 2 * pad - ks) / stride + 1; }

// geometry of a conv_wgrad_c16_kernel launch
struct WgC16Args {
    int N, Cin, H, Wd, Cout, Ho, Wo, pad;
    int TH, PW4, tiles_y, ntiles, nslab;
    int XR, XP, nkb, nW;
};

}  // namespace aq

// ---- aq_conv_c16.hip ------------------------------------------------------------------------------------------
// forward (stride 1 | 2) / input gradient (stride 1, and stride 2 by parity classes) of 3x3 layers with 16-channel
// multiples in and at least 48 out; *taken = false: not one of its shapes, nothing launched
int aq_c16_launch_fwd(aq::ConvArgs a, int N, hipStream_t st, const char *who, bool *taken);
int aq_c16_launch_dgrad(aq::ConvArgs a, int N, hipStream_t st, const char *who, bool *taken);

// ---- aq_conv_1x1.hip ------------------------------------------------------------------------------------------
// forward / input gradient of 1x1 layers without padding (stride 1 | 2) whose channel counts are multiples of 16;
// *taken = false: not one of its shapes, nothing launched
int aq_c1_launch_fwd(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken);
int aq_c1_launch_dgrad(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken);

// ---- aq_conv_c16.hip (continued) ------------------------------------------------------------------------------
// weight gradient of 3x3 (input channels a multiple of 16) and 1x1 (a multiple of 64) layers with output channels a
// multiple of 16: plan (use = false: not one of its shapes) and launch for a group of layers
struct WgC16Plan {
    bool use;
    int NG, CW;
    aq::WgC16Args a;
    dim3 grid;
    size_t lds, ws_bytes;
};
WgC16Plan wgrad_c16_plan(int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad, int nlayers = 1);
int wgrad_c16_launch(const WgC16Plan &cp, const aq::WgradGroup &g, int ks, int stride, hipStream_t st);

// ---- aq_conv_stem.hip -----------------------------------------------------------------------------------------
// forward of the image stem (RGB, 7x7, stride 2, output channels a multiple of 64, rows of at most 112 output pixels);
// *taken = false: not one of its shapes, nothing launched
int aq_stem_launch_fwd(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken);
// its weight gradient (rows of a multiple of 4 output pixels, dy 16-byte aligned): plan (use = false: not one of its
// shapes) and launch of the slab kernel
struct WgStemPlan {
    bool use;
    int nslab;
    size_t ws_bytes;
};
WgStemPlan wgrad_stem_plan(const float *dy, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad);
int wgrad_stem_launch(const float *x, const float *dy, float *part, int N, int Cin, int H, int Wd, int Cout, int ks, int stride, int pad,
                      hipStream_t st);

// ---- aq_conv_wino.hip -----------------------------------------------------------------------------------------
// forward / input gradient of 3x3, stride 1, padding 1 layers with at least 64 channels in and out (multiples of 16) by
// the F(2x2, 3x3) minimal-filtering form; *taken = false: not one of its shapes, nothing launched
int aq_wino_launch_fwd(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken);
int aq_wino_launch_dgrad(const aq::ConvArgs &a, int N, hipStream_t st, const char *who, bool *taken);
