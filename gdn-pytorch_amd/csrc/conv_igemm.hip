// Implicit-GEMM convolution for gfx950 (MI355X), exact fp32 on the matrix cores.
//
//   out[m][n] = sum_{tap, c} in[pix(m) (+) tap][c] * w[tap][n][c]
//
// One kernel serves every convolution-shaped op on the GDN hot path:
//   * Conv2d forward, stride 1/2, zero or reflection padding (im2col-free: the
//     A tile of a k-step is gathered straight from the NHWC activation, one
//     filter tap and one 32-channel slab at a time, 128-B coalesced per pixel);
//   * ConvTranspose2d forward and strided-conv data gradients as `stride^2`
//     output phases in ONE launch (grid.y = phase), each phase a dense
//     stride-1 gather over its own tap subset -- no zero-insertion;
//   * concat-free 1x1 conv: reduction channels [0,C1) come from x, the rest from x2;
//   * epilogue fusions: per-channel sum / sum-of-squares partials for train-mode
//     BatchNorm, residual/gradient accumulation (addsrc), tanh.
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU).  A wave
// owns TM x TN tiles of 32x32; a k-step is 32 reduction channels.  The MFMA
// reduces over k in any order, so lane-half h takes channels [16h,16h+16) of
// the slab: every fragment load is a 16-byte ds_read_b128 (4 k-substeps per
// read) from an LDS image with a 36-float row pitch (conflict-free for the
// 16-lane ds_read_b128 groups).  Global -> register -> LDS staging is split
// (issue loads for k-step i+1, run the MFMAs of k-step i, then write LDS) so
// HBM/L2 latency hides under the 2-4k MFMA cycles of a k-step.
#include "common.h"
#include "up2x.h"

#define MAX_TAPS 81
#define MAX_PHASE 4
#define KC_MIN 32          // smallest reduction slab (also the scalar-gather slab)

struct IgemmPhase { int Ho, Wo, oy0, ox0, tap_begin, tap_end; };

struct IgemmParams {
    const float* x; const float* x2; const float* w; float* y; const float* addsrc; float* stats;
    const float* ep_scale; const float* ep_shift;   // optional per-channel affine of the epilogue (eval-mode BN fold)
    int B, Hi, Wi, C1, C2, ldx1, ldx2;
    int plan_B, plan_cus;            // gdn_plan_batch() / gdn_plan_cus() of the geometry: what the host-side plans are made for
    int N, ldy, ld_add, Hy, Wy, osy, osx;
    int stride, pad_mode, act, nphase;
    int Cred, w_tap_stride;
    int grid_m, grid_n;
    unsigned long long x_bytes, x2_bytes;  // tensor extents; descriptors are re-based per workgroup (32-bit offsets)
    unsigned w_bytes;
    int kc;                                // reduction slab per k-step (32 or 64)
    int bf16;                              // 1: x/x2/w/y/addsrc hold bf16 (fp32 accumulate, fp32 stats/partials)
    int ksplit;                            // split-K over taps (grid.z); >1: raw partials go to `part`
    float* part;                           // [ksplit][npix_out][N]  (conv_ring_bf16: the tail ranges' slabs [parts (x2)][tail pixels][N])
    long long npix_out;
    int ring_main, ring_sp;                // conv_ring_bf16 with ksplit = parts > 1: units of the full rounds, stages per tail range
    int ring_notail;                       // tile_cfg & 0x800 ("single stage"): the last round's units stay whole (A/B measurements, tests)
    // conv_ring_bf16 as a data gradient whose output is the FINAL gradient of z = [relu](BN_train(bnb_y)): `stats` then receives that
    // BatchNorm's backward partials (sum dz, sum dz * xhat per slot) instead of sum / sum of squares (bnb_co = [scale, shift, mean, invstd][N])
    const void* bnb_y; const float* bnb_co; int ld_bnb, bnb_relu;
    IgemmPhase ph[MAX_PHASE];
    short tdy[MAX_TAPS], tdx[MAX_TAPS], twi[MAX_TAPS];
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SCALAR, int KC>
__global__ __launch_bounds__(256) void conv_igemm_f32(const IgemmParams p) {
    constexpr int LDS_LD = KC + 4;               // row pitch: conflict-free ds_read_b128 for 36 and 68 floats
    constexpr int TPR = KC / 4;                  // threads per staged row (16 B each)
    constexpr int RPP = 256 / TPR;               // rows per staging pass
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int A_FLOATS = BM * LDS_LD, B_FLOATS = BN * LDS_LD;
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS + 3 * BM];
    float* As = smem;
    float* Bs = smem + A_FLOATS;
    int* row_pix = reinterpret_cast<int*>(smem + A_FLOATS + B_FLOATS);
    int* row_out = row_pix + BM;
    int* row_yx = row_out + BM;   // iy0 | ix0 << 16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ph_i = blockIdx.y;
    IgemmPhase ph = p.ph[ph_i];
    const int kz = blockIdx.z;
    if (p.ksplit > 1) {          // split-K: this workgroup reduces a contiguous share of the phase's taps
        const int tb = ph.tap_begin, nt_all = ph.tap_end - ph.tap_begin;
        ph.tap_begin = tb + (nt_all * kz) / p.ksplit;
        ph.tap_end = tb + (nt_all * (kz + 1)) / p.ksplit;
    }

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (its L2).  Each XCD gets a contiguous BAND of
    // M-tiles (so the tiles above/below a tile -- which re-read its rows through the filter's halo -- are
    // resident on the same L2 at about the same time: a 9x9 layer otherwise fetches every input row nine
    // times from beyond L2), and the N-tiles of one M-tile stay adjacent on that XCD.
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    const int per_xcd = (p.grid_m + 7) >> 3;
    const int nt = q % p.grid_n, mt = xcd * per_xcd + q / p.grid_n;
    if (q / p.grid_n >= per_xcd || mt >= p.grid_m) return;
    const int M = p.B * ph.Ho * ph.Wo;
    const int m0 = mt * BM, n0 = nt * BN;
    const int slot = ph_i * p.grid_m + mt;
    if (m0 >= M) {
        if (p.ksplit == 1 && p.stats && tid < BN && n0 + tid < p.N) {
            p.stats[((size_t)slot * 2 + 0) * p.N + n0 + tid] = 0.f;
            p.stats[((size_t)slot * 2 + 1) * p.N + n0 + tid] = 0.f;
        }
        return;
    }

    // Buffer descriptors address 32 bits, activations can exceed 4 GiB (B=64 at 256x832): base every
    // descriptor at the first image this workgroup touches (a 64-row tile spans at most two images).
    const int b_first = m0 / (ph.Ho * ph.Wo);
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        if (m < M) {
            const int ox = m % ph.Wo, t = m / ph.Wo, oy = t % ph.Ho, b = t / ph.Ho;
            row_pix[r] = (b - b_first) * p.Hi * p.Wi;      // relative to this workgroup's first image
            row_yx[r] = (oy * p.stride) | ((ox * p.stride) << 16);
            row_out[r] = (b * p.Hy + oy * p.osy + ph.oy0) * p.Wy + ox * p.osx + ph.ox0;
        } else {
            row_pix[r] = -1; row_yx[r] = 0; row_out[r] = -1;
        }
    }
    __syncthreads();

    const int ntap = ph.tap_end - ph.tap_begin;
    const int nchunks = p.Cred / KC;     // host guarantees divisibility for the chosen KC
    const int Ktot = ntap * p.Cred;
    const int nk = SCALAR ? (Ktot + KC - 1) / KC : ntap * nchunks;

    f32x16 acc[TM][TN], part[TM][TN], zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero16;

    // ---------------- staging registers ----------------
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;     // VEC: TPR threads x 16 B per row
    constexpr int A_KP = 256 / BM > 0 ? 256 / BM : 1, A_E = KC / A_KP;   // SCALAR
    constexpr int B_KP = 256 / BN > 0 ? 256 / BN : 1, B_E = KC / B_KP;
    f32x4 ra[SCALAR ? 1 : A_PASSES], rb[SCALAR ? 1 : B_PASSES];
    float sa[SCALAR ? A_E : 1], sb[SCALAR ? B_E : 1];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Loop-invariant address pieces (VEC path).  The k-steps walk taps in the outer loop and
    // 32-channel slabs in the inner one; a row's gather position depends on the tap only, so the
    // bounds / reflection logic runs once per tap, not once per k-step, and B rows never change.
    // All tile loads are raw buffer loads: the per-lane part is a 32-bit byte offset, the per-k-step
    // part (tap, slab) rides in the scalar offset, and rows that must read as zero (padding, M/N
    // tails) simply carry an out-of-range offset -- the hardware range check returns 0, so there is
    // no branch, no select and no 64-bit address arithmetic in the loop.  On this chip the fp32
    // MFMA shares the vector ALUs (measured: time/MFMA ~ 64 + 4 * VALU-per-MFMA cycles), so every
    // VALU instruction removed from the k-step is MFMA time won back.
    int aoff[SCALAR ? 1 : A_PASSES];               // gathered pixel index of this thread's rows, -1 = zero
    unsigned boff[SCALAR ? 1 : B_PASSES];          // byte offset of (row n, 16-B lane chunk), OOB = row beyond N
    int tl_n = 0, cc_n = 0, wi_n = 0;              // prefetch cursor: tap, slab, weight tap index
    const unsigned lane_b = (unsigned)(tid % TPR) * 16u;
    const int row_in_pass = tid / TPR;
    const unsigned long long img1 = (unsigned long long)p.Hi * p.Wi * p.ldx1 * 4ull * b_first;
    const unsigned long long img2 = (unsigned long long)p.Hi * p.Wi * p.ldx2 * 4ull * b_first;
    const unsigned long long rem1 = p.x_bytes - img1, rem2 = p.x2 ? p.x2_bytes - img2 : rem1;
    const unsigned long long cap = 0xFF000000ull;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + img1), 0, (int)(unsigned)(rem1 < cap ? rem1 : cap), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.x2 ? reinterpret_cast<const char*>(p.x2) + img2 : reinterpret_cast<const char*>(p.x) + img1), 0,
        (int)(unsigned)(rem2 < cap ? rem2 : cap), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    if constexpr (!SCALAR) {
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const int n = n0 + ps * RPP + row_in_pass;
            boff[ps] = n < p.N ? (unsigned)(n * p.Cred) * 4u + lane_b : 0xFFFFFF00u;
        }
    }
    auto set_tap = [&](int tl) {
        if constexpr (!SCALAR) {
            const int t = ph.tap_begin + tl;
            const int dy = p.tdy[t], dx = p.tdx[t];
            wi_n = p.twi[t];
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int r = ps * RPP + row_in_pass;
                const int base = row_pix[r], yx = row_yx[r];
                int iy = (yx & 0xffff) + dy, ix = (yx >> 16) + dx;
                if (p.pad_mode == 1) { iy = reflect_idx(iy, p.Hi); ix = reflect_idx(ix, p.Wi); }
                const bool ok = base >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                aoff[ps] = ok ? base + iy * p.Wi + ix : -1;       // -1 * ld*4 wraps far out of range
            }
        }
    };

    auto gload = [&](int ks) {
        if constexpr (!SCALAR) {
            if (cc_n == 0) set_tap(tl_n);
            const int ci0 = cc_n * KC;
            const bool first = ci0 < p.C1;
            const unsigned ld4 = (unsigned)(first ? p.ldx1 : p.ldx2) * 4u;
            const int soff = (first ? ci0 : ci0 - p.C1) * 4;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const unsigned vo = (unsigned)aoff[ps] * ld4 + lane_b;
                ra[ps] = __builtin_bit_cast(f32x4, first ? __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, soff, 0)
                                                         : __builtin_amdgcn_raw_buffer_load_b128(rs_x2, vo, soff, 0));
            }
            const int soff_w = (wi_n * p.w_tap_stride + ci0) * 4;
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps)
                rb[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[ps], soff_w, 0));
            if (++cc_n == nchunks) { cc_n = 0; ++tl_n; }
        } else {
            {
                const int r = tid % BM, kp = tid / BM;
                const int base = row_pix[r], yx = row_yx[r];
#pragma unroll
                for (int e = 0; e < A_E; ++e) {
                    const int kk = ks * KC + kp * A_E + e;
                    float v = 0.f;
                    if (kk < Ktot && base >= 0 && kp < A_KP) {
                        const int tl = kk / p.Cred, ci = kk - tl * p.Cred, t = ph.tap_begin + tl;
                        int iy = (yx & 0xffff) + p.tdy[t], ix = (yx >> 16) + p.tdx[t];
                        bool ok = true;
                        if (p.pad_mode == 1) { iy = reflect_idx(iy, p.Hi); ix = reflect_idx(ix, p.Wi); }
                        ok = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
                        if (ok) v = p.x[((size_t)b_first * p.Hi * p.Wi + (size_t)(base + iy * p.Wi + ix)) * p.ldx1 + ci];
                    }
                    sa[e] = v;
                }
            }
            {
                const int n = n0 + tid % BN, kp = tid / BN;
#pragma unroll
                for (int e = 0; e < B_E; ++e) {
                    const int kk = ks * KC + kp * B_E + e;
                    float v = 0.f;
                    if (kk < Ktot && n < p.N && kp < B_KP) {
                        const int tl = kk / p.Cred, ci = kk - tl * p.Cred, t = ph.tap_begin + tl;
                        v = p.w[(size_t)p.twi[t] * p.w_tap_stride + (size_t)n * p.Cred + ci];
                    }
                    sb[e] = v;
                }
            }
        }
    };

    auto lds_store = [&]() {
        if constexpr (!SCALAR) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps)
                *reinterpret_cast<f32x4*>(&As[(ps * RPP + row_in_pass) * LDS_LD + (tid % TPR) * 4]) = ra[ps];
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps)
                *reinterpret_cast<f32x4*>(&Bs[(ps * RPP + row_in_pass) * LDS_LD + (tid % TPR) * 4]) = rb[ps];
        } else {
            if (tid / BM < A_KP) {
#pragma unroll
                for (int e = 0; e < A_E; ++e) As[(tid % BM) * LDS_LD + (tid / BM) * A_E + e] = sa[e];
            }
            if (tid / BN < B_KP) {
#pragma unroll
                for (int e = 0; e < B_E; ++e) Bs[(tid % BN) * LDS_LD + (tid / BN) * B_E + e] = sb[e];
            }
        }
    };

    const int a_off = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * (KC / 2);
    const int b_off = (wn * TN * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * (KC / 2);

    gload(0);
    lds_store();
    __syncthreads();
    // Two-level summation: the MFMA is a strictly k-ordered fp32 fma chain, so one chain over
    // K = taps*Cin (up to 6.4k terms) would carry sqrt(K) rounding growth.  FOLD k-steps (128 terms)
    // run in a fresh chain (C = 0 on the first MFMA) and are then folded into the running total.
    constexpr int FOLD = 128 / KC;
    for (int ks0 = 0; ks0 < nk; ks0 += FOLD) {
#pragma unroll
        for (int j = 0; j < FOLD; ++j) {
            const int ks = ks0 + j;
            if (ks >= nk) break;
            if (ks + 1 < nk) gload(ks + 1);
#pragma unroll
            for (int g = 0; g < KC / 8; ++g) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[a_off + i * 32 * LDS_LD + g * 4]);
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bf[jn] = *reinterpret_cast<const f32x4*>(&Bs[b_off + jn * 32 * LDS_LD + g * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jn = 0; jn < TN; ++jn)
                            part[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                af[i][e], bf[jn][e], (j == 0 && g == 0 && e == 0) ? zero16 : part[i][jn], 0, 0, 0);
            }
            __syncthreads();
            if (ks + 1 < nk) {
                lds_store();
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] += part[i][jn];
    }

    // ---------------- epilogue ----------------
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const int col_l = lane & 31, rsh = 4 * (lane >> 5);
    if (p.ksplit > 1) {          // raw partial sums; splitk_combine_kernel finishes (sum, addsrc, act, BN stats)
        float* pp = p.part + (size_t)kz * p.npix_out * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
                const int op = row_out[row];
                if (op < 0) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * TN * 32 + j * 32 + col_l;
                    if (n < p.N) pp[(size_t)op * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    if (p.stats) {
        float* red = As;  // [WAVES_M][BN][2], safe: all waves are past their last LDS read
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int c = wn * TN * 32 + j * 32 + lane;
                red[(wm * BN + c) * 2 + 0] = s1;
                red[(wm * BN + c) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int wmi = 0; wmi < WAVES_M; ++wmi) { s1 += red[(wmi * BN + tid) * 2]; s2 += red[(wmi * BN + tid) * 2 + 1]; }
            p.stats[((size_t)slot * 2 + 0) * p.N + n0 + tid] = s1;
            p.stats[((size_t)slot * 2 + 1) * p.N + n0 + tid] = s2;
        }
    }
    const bool has_affine = p.ep_scale != nullptr;
    float es[TN], et[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + col_l;
        es[j] = (has_affine && n < p.N) ? p.ep_scale[n] : 1.f;
        et[j] = (has_affine && n < p.N) ? p.ep_shift[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
            const int op = row_out[row];
            if (op < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + col_l;
                if (n < p.N) {
                    float v = acc[i][j][r];
                    if (has_affine) v = v * es[j] + et[j];
                    if (p.act & GDN_ACT_RELU) v = fmaxf(v, 0.f);
                    if (p.addsrc) v += p.addsrc[(size_t)op * p.ld_add + n];
                    if (p.act & GDN_ACT_TANH) v = tanhf(v);
                    p.y[(size_t)op * p.ldy + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// bf16 variant (BASELINE configs[2]): bf16 activations/weights, v_mfma_f32_32x32x16_bf16, fp32
// accumulate, fp32 BatchNorm statistics.  Same byte geometry as the fp32 kernel -- a k-step row is
// 128 bytes (64 channels), 8 lanes x 16 B per row, 144-byte LDS pitch, lane-half h owns bytes
// [64h, 64h+64) of the row -- so loader, LDS image and tile/phase logic are shared; one 16-byte
// fragment read now feeds ONE MFMA (8 bf16 k-values) instead of four.  The MFMA is 16x faster than
// the fp32 one, so the tile is 128x128 (64 MFMAs per wave per k-step) to keep it fed.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return f32_to_bf16_h(f); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// (four waves per SIMD: the 128 x 128 form then fits 128 registers without a spill instead of 88 + 64 accumulation registers at
//  three waves -- this single-image, two-barrier pipeline lives on occupancy: +2 % over the strided / transposed / 1x1 layers,
//  while a second LDS image at half the workgroups per CU cost 20 %, profiles/README.md)
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 4) void conv_igemm_bf16(const IgemmParams p) {
    constexpr int KC = 64, LDS_LD = 36, RPP = 32;        // 64 bf16 = 128 B per row, 144-B pitch
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int A_FLOATS = BM * LDS_LD, B_FLOATS = BN * LDS_LD;
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS + 3 * BM];
    float* As = smem;
    float* Bs = smem + A_FLOATS;
    int* row_pix = reinterpret_cast<int*>(smem + A_FLOATS + B_FLOATS);
    int* row_out = row_pix + BM;
    int* row_yx = row_out + BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ph_i = blockIdx.y;
    IgemmPhase ph = p.ph[ph_i];
    const int kz = blockIdx.z;
    if (p.ksplit > 1) {
        const int tb = ph.tap_begin, nt_all = ph.tap_end - ph.tap_begin;
        ph.tap_begin = tb + (nt_all * kz) / p.ksplit;
        ph.tap_end = tb + (nt_all * (kz + 1)) / p.ksplit;
    }
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    const int per_xcd = (p.grid_m + 7) >> 3;
    const int nt = q % p.grid_n, mt = xcd * per_xcd + q / p.grid_n;
    if (q / p.grid_n >= per_xcd || mt >= p.grid_m) return;
    const int M = p.B * ph.Ho * ph.Wo;
    const int m0 = mt * BM, n0 = nt * BN;
    const int slot = ph_i * p.grid_m + mt;
    if (m0 >= M) {
        if (p.ksplit == 1 && p.stats && tid < BN && n0 + tid < p.N) {
            p.stats[((size_t)slot * 2 + 0) * p.N + n0 + tid] = 0.f;
            p.stats[((size_t)slot * 2 + 1) * p.N + n0 + tid] = 0.f;
        }
        return;
    }
    const int b_first = m0 / (ph.Ho * ph.Wo);
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        if (m < M) {
            const int ox = m % ph.Wo, t = m / ph.Wo, oy = t % ph.Ho, b = t / ph.Ho;
            row_pix[r] = (b - b_first) * p.Hi * p.Wi;
            row_yx[r] = (oy * p.stride) | ((ox * p.stride) << 16);
            row_out[r] = (b * p.Hy + oy * p.osy + ph.oy0) * p.Wy + ox * p.osx + ph.ox0;
        } else {
            row_pix[r] = -1; row_yx[r] = 0; row_out[r] = -1;
        }
    }
    __syncthreads();

    const int ntap = ph.tap_end - ph.tap_begin;
    const int nchunks = p.Cred / KC;
    const int nk = ntap * nchunks;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    f32x4 ra[A_PASSES], rb[B_PASSES];
    int aoff[A_PASSES];
    unsigned boff[B_PASSES];
    int tl_n = 0, cc_n = 0, wi_n = 0;
    const unsigned lane_b = (unsigned)(tid & 7) * 16u;
    const int row_in_pass = tid >> 3;
    const unsigned long long img1 = (unsigned long long)p.Hi * p.Wi * p.ldx1 * 2ull * b_first;
    const unsigned long long img2 = (unsigned long long)p.Hi * p.Wi * p.ldx2 * 2ull * b_first;
    const unsigned long long rem1 = p.x_bytes - img1, rem2 = p.x2 ? p.x2_bytes - img2 : rem1;
    const unsigned long long cap = 0xFF000000ull;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + img1), 0, (int)(unsigned)(rem1 < cap ? rem1 : cap), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.x2 ? reinterpret_cast<const char*>(p.x2) + img2 : reinterpret_cast<const char*>(p.x) + img1), 0,
        (int)(unsigned)(rem2 < cap ? rem2 : cap), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
        const int n = n0 + ps * RPP + row_in_pass;
        boff[ps] = n < p.N ? (unsigned)(n * p.Cred) * 2u + lane_b : 0xFFFFFF00u;
    }
    auto set_tap = [&](int tl) {
        const int t = ph.tap_begin + tl;
        const int dy = p.tdy[t], dx = p.tdx[t];
        wi_n = p.twi[t];
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int r = ps * RPP + row_in_pass;
            const int base = row_pix[r], yx = row_yx[r];
            int iy = (yx & 0xffff) + dy, ix = (yx >> 16) + dx;
            if (p.pad_mode == 1) { iy = reflect_idx(iy, p.Hi); ix = reflect_idx(ix, p.Wi); }
            const bool ok = base >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            aoff[ps] = ok ? base + iy * p.Wi + ix : -1;
        }
    };
    auto gload = [&]() {
        if (cc_n == 0) set_tap(tl_n);
        const int ci0 = cc_n * KC;
        const bool first = ci0 < p.C1;
        const unsigned ld2 = (unsigned)(first ? p.ldx1 : p.ldx2) * 2u;
        const int soff = (first ? ci0 : ci0 - p.C1) * 2;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const unsigned vo = (unsigned)aoff[ps] * ld2 + lane_b;
            ra[ps] = __builtin_bit_cast(f32x4, first ? __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, soff, 0)
                                                     : __builtin_amdgcn_raw_buffer_load_b128(rs_x2, vo, soff, 0));
        }
        const int soff_w = (wi_n * p.w_tap_stride + ci0) * 2;
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps)
            rb[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[ps], soff_w, 0));
        if (++cc_n == nchunks) { cc_n = 0; ++tl_n; }
    };
    auto lds_store = [&]() {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
            *reinterpret_cast<f32x4*>(&As[(ps * RPP + row_in_pass) * LDS_LD + (tid & 7) * 4]) = ra[ps];
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps)
            *reinterpret_cast<f32x4*>(&Bs[(ps * RPP + row_in_pass) * LDS_LD + (tid & 7) * 4]) = rb[ps];
    };
    const int a_off = (wm * TM * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 16;
    const int b_off = (wn * TN * 32 + (lane & 31)) * LDS_LD + (lane >> 5) * 16;

    gload();
    lds_store();
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) gload();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[a_off + i * 32 * LDS_LD + g * 4]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[b_off + j * 32 * LDS_LD + g * 4]));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (ks + 1 < nk) {
            lds_store();
            __syncthreads();
        }
    }

    const int col_l = lane & 31, rsh = 4 * (lane >> 5);
    if (p.ksplit > 1) {
        float* pp = p.part + (size_t)kz * p.npix_out * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
                const int op = row_out[row];
                if (op < 0) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * TN * 32 + j * 32 + col_l;
                    if (n < p.N) pp[(size_t)op * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    if (p.stats) {
        float* red = As;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int c = wn * TN * 32 + j * 32 + lane;
                red[(wm * BN + c) * 2 + 0] = s1;
                red[(wm * BN + c) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int wmi = 0; wmi < WAVES_M; ++wmi) { s1 += red[(wmi * BN + tid) * 2]; s2 += red[(wmi * BN + tid) * 2 + 1]; }
            p.stats[((size_t)slot * 2 + 0) * p.N + n0 + tid] = s1;
            p.stats[((size_t)slot * 2 + 1) * p.N + n0 + tid] = s2;
        }
    }
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* ad = reinterpret_cast<const unsigned short*>(p.addsrc);
    const bool has_affine = p.ep_scale != nullptr;
    float es[TN], et[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + col_l;
        es[j] = (has_affine && n < p.N) ? p.ep_scale[n] : 1.f;
        et[j] = (has_affine && n < p.N) ? p.ep_shift[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
            const int op = row_out[row];
            if (op < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + col_l;
                if (n < p.N) {
                    float v = acc[i][j][r];
                    if (has_affine) v = v * es[j] + et[j];
                    if (p.act & GDN_ACT_RELU) v = fmaxf(v, 0.f);
                    if (ad) v += bf16_to_f32(ad[(size_t)op * p.ld_add + n]);
                    if (p.act & GDN_ACT_TANH) v = tanhf(v);
                    yo[(size_t)op * p.ldy + n] = f32_to_bf16(v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Row-patch bf16 kernel for the stride-1 layers with k >= 3 (the residual blocks: 80 % of the FLOPs).
// At bf16 MFMA speed the tap-major kernel above is bound by L2->LDS traffic: a 128x64 tile pulls 24.5 KB per
// 1 MFLOP, ~14.5 TB/s at the measured rate on the 9x9 layers.  Here a workgroup owns 256 consecutive output
// pixels (flattened over image rows) and, per filter ROW and 64-channel slab, stages the input rows those
// pixels need -- each with its k-1 halo, <= 288 positions -- in LDS ONCE; the k taps of the row are the same
// image read at a shifted row address, so activations are fetched k times instead of k*k times and only the
// weight tile (BN x 64 channels per tap) streams.  Traffic per FLOP drops 2.4x (3x3) to 4x (9x9).
// Same fragment byte geometry, epilogues (BN statistics, residual, tanh) and XCD mapping as conv_igemm_bf16.
// ---------------------------------------------------------------------------
#define RP_BM 256
#define RP_PMAX 288          // staged positions: 256 pixels + (k-1) halo per touched image row
#define RP_NRMAX 12          // image rows a 256-pixel tile may touch (W >= 26)
#define RP_KMAX 9

// (three waves per SIMD requested: left alone the compiler takes 228 registers for the 256x64 form = two workgroups per
//  CU; within 170 (166 used, no spills) three fit -- 52 KB of LDS each, 157 of 160 KB -- and the 9x9 64-channel layers go
//  from 790-855 to 990-1040 TFLOP/s.  The 256x128 form cannot fit 170 and spills; it is kept for tuning runs only.)
template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 3) void conv_rowpatch_bf16(const IgemmParams p) {
    constexpr int BM = RP_BM, KC = 64, PITCH = 144;                  // bytes per staged row (128 data + 16 pad)
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int AV = RP_PMAX * 8 / 256, BV = BN * 8 / 256;         // 16-B loads per thread: patch, weight tile
    __shared__ __attribute__((aligned(16))) unsigned char As[RP_PMAX * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * PITCH];
    __shared__ int row_out[BM];
    __shared__ int rowoff[RP_KMAX][RP_NRMAX];        // byte offset of input row (filter row, patch row) or -1
    __shared__ int rbase[RP_NRMAX + 1];              // first patch position of each touched image row
    __shared__ int rxlo[RP_NRMAX];                   // first output column of the tile in that row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const IgemmPhase ph = p.ph[0];
    const int bid = blockIdx.x, xcd = bid & 7, q8 = bid >> 3;
    const int per_xcd = (p.grid_m + 7) >> 3;
    const int nt = q8 % p.grid_n, mt = xcd * per_xcd + q8 / p.grid_n;
    if (q8 / p.grid_n >= per_xcd || mt >= p.grid_m) return;
    const int Wo = ph.Wo, Ho = ph.Ho;
    const int M = p.B * Ho * Wo;
    const int m0 = mt * BM, n0 = nt * BN;
    const int slot = mt;
    const int ntap = ph.tap_end - ph.tap_begin;
    int kw = 1;                                       // taps per filter row (consecutive taps sharing tdy)
    while (kw < ntap && p.tdy[ph.tap_begin + kw] == p.tdy[ph.tap_begin]) ++kw;
    const int kh = ntap / kw;
    int dxmin = p.tdx[ph.tap_begin];
    for (int t = 1; t < kw; ++t) dxmin = min(dxmin, (int)p.tdx[ph.tap_begin + t]);
    const int mlast = min(m0 + BM, M) - 1;
    const int row0 = m0 / Wo, nrows = mlast / Wo - row0 + 1;
    const int b_first = row0 / Ho;

    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        if (m < M) {
            const int ox = m % Wo, t = m / Wo, oy = t % Ho, b = t / Ho;
            row_out[r] = (b * p.Hy + oy * p.osy + ph.oy0) * p.Wy + ox * p.osx + ph.ox0;
        } else row_out[r] = -1;
    }
    if (tid <= nrows) {
        // pixels of the tile that precede row j: 0 for j = 0, else (Wo - xlo0) + (j-1)*Wo
        const int xlo0 = m0 - row0 * Wo;
        const int before = tid == 0 ? 0 : (Wo - xlo0) + (tid - 1) * Wo;
        rbase[tid] = (tid == nrows ? min(BM, M - m0) : before) + tid * (kw - 1);
        if (tid < nrows) rxlo[tid] = tid == 0 ? xlo0 : 0;
    }
    for (int i = tid; i < kh * RP_NRMAX; i += 256) {
        const int ky = i / RP_NRMAX, j = i - ky * RP_NRMAX;
        int off = -1;
        if (j < nrows) {
            const int t = row0 + j, oy = t % Ho, b = t / Ho;
            int iy = oy * p.stride + p.tdy[ph.tap_begin + ky * kw];
            if (p.pad_mode == 1) iy = reflect_idx(iy, p.Hi);
            if ((unsigned)iy < (unsigned)p.Hi) off = (((b - b_first) * p.Hi + iy) * p.Wi) * p.ldx1 * 2;
        }
        rowoff[ky][j] = off;
    }
    __syncthreads();

    // ---- staging constants ----
    const unsigned OOB = 0xFFFFFF00u;
    const unsigned lane_b = (unsigned)(tid & 7) * 16u;
    const int pos0 = tid >> 3;
    const int npatch = rbase[nrows];
    unsigned pk_x[AV]; int pk_j[AV];
#pragma unroll
    for (int e = 0; e < AV; ++e) {
        const int q = pos0 + 32 * e;
        int j = 0;
        for (int jj = 1; jj < nrows; ++jj) j += (q >= rbase[jj]) ? 1 : 0;
        int ix = rxlo[j] + dxmin + (q - rbase[j]);
        if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wi);
        const bool ok = q < npatch && (unsigned)ix < (unsigned)p.Wi;
        pk_x[e] = ok ? (unsigned)(ix * p.ldx1) * 2u + lane_b : OOB;
        pk_j[e] = j;
    }
    const unsigned long long img1 = (unsigned long long)p.Hi * p.Wi * p.ldx1 * 2ull * b_first;
    const unsigned long long rem1 = p.x_bytes - img1, cap = 0xFF000000ull;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + img1), 0, (int)(unsigned)(rem1 < cap ? rem1 : cap), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    unsigned boff[BV];
#pragma unroll
    for (int e = 0; e < BV; ++e) {
        const int n = n0 + e * 32 + pos0;
        boff[e] = n < p.N ? (unsigned)(n * p.Cred) * 2u + lane_b : OOB;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A fragment rows: pixel r of the tile sits at patch row r + j(r)*(kw-1)
    unsigned arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * TM * 32 + i * 32 + (lane & 31);
        int j = 0;
        const int m = min(m0 + r, mlast);
        j = m / Wo - row0;
        arow[i] = (unsigned)(min(r, mlast - m0) + j * (kw - 1)) * PITCH + (unsigned)(lane >> 5) * 64u;
    }
    const unsigned brow = (unsigned)(wn * TN * 32 + (lane & 31)) * PITCH + (unsigned)(lane >> 5) * 64u;

    const int nchunks = p.Cred / KC;
    const int nstage = kh * nchunks;                 // (filter row, channel slab)
    f32x4 ra[AV], rb[BV];
    auto load_a = [&](int stage) {
        const int ky = stage / nchunks, ci0 = (stage - ky * nchunks) * KC;
#pragma unroll
        for (int e = 0; e < AV; ++e) {
            const int ro = rowoff[ky][pk_j[e]];
            const unsigned vo = (ro >= 0 && pk_x[e] != OOB) ? (unsigned)ro + pk_x[e] : OOB;
            ra[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, ci0 * 2, 0));
        }
    };
    auto load_b = [&](int stage, int kx) {
        const int ky = stage / nchunks, ci0 = (stage - ky * nchunks) * KC;
        const int wi = p.twi[ph.tap_begin + ky * kw + kx];
        const int soff = (wi * p.w_tap_stride + ci0) * 2;
#pragma unroll
        for (int e = 0; e < BV; ++e)
            rb[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[e], soff, 0));
    };
    auto store_a = [&]() {
#pragma unroll
        for (int e = 0; e < AV; ++e)
            *reinterpret_cast<f32x4*>(&As[(pos0 + 32 * e) * PITCH + (tid & 7) * 16]) = ra[e];
    };
    auto store_b = [&]() {
#pragma unroll
        for (int e = 0; e < BV; ++e)
            *reinterpret_cast<f32x4*>(&Bs[(pos0 + 32 * e) * PITCH + (tid & 7) * 16]) = rb[e];
    };

    // Pipeline: a tap step is only 16-32 MFMAs (0.5-1k cycles) against ~2.5k cycles of load latency, so loads run
    // ahead of their use: the next stage's patch is requested at the FIRST tap of the current stage (its registers are
    // idle until the stage ends anyway) and weight tiles two taps ahead (two register sets, alternating).
    const int nsteps = nstage * kw;
    auto load_b_step = [&](int step, f32x4* dst) {
        const int stage = step / kw, kx = step - stage * kw;
        const int ky = stage / nchunks, ci0 = (stage - ky * nchunks) * KC;
        const int wi = p.twi[ph.tap_begin + ky * kw + kx];
        const int soff = (wi * p.w_tap_stride + ci0) * 2;
#pragma unroll
        for (int e = 0; e < BV; ++e)
            dst[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[e], soff, 0));
    };
    auto store_b_from = [&](const f32x4* src) {
#pragma unroll
        for (int e = 0; e < BV; ++e)
            *reinterpret_cast<f32x4*>(&Bs[(pos0 + 32 * e) * PITCH + (tid & 7) * 16]) = src[e];
    };
    f32x4 rb2[BV];
    load_a(0);
    load_b_step(0, rb);
    store_a();
    store_b_from(rb);
    if (nsteps > 1) load_b_step(1, rb);
    if (nsteps > 2) load_b_step(2, rb2);
    __syncthreads();
    auto step_body = [&](int step, f32x4* cur_next, f32x4* refill) {
        // cur_next holds the weight tile of step+1 (already requested); refill receives step+3's
        const int stage = step / kw, kx = step - stage * kw;
        if (kx == 0 && stage + 1 < nstage) load_a(stage + 1);
        const unsigned ashift = (unsigned)(p.tdx[ph.tap_begin + (stage / nchunks) * kw + kx] - dxmin) * PITCH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&As[arow[i] + ashift + g * 16]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Bs[brow + j * 32 * PITCH + g * 16]));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (step + 1 < nsteps) {
            if (kx + 1 == kw) store_a();
            store_b_from(cur_next);
            if (step + 3 < nsteps) load_b_step(step + 3, refill);
            __syncthreads();
        }
    };
    for (int step = 0; step < nsteps; step += 2) {
        step_body(step, rb, rb);              // consumes rb (step+1), refills rb with step+3
        if (step + 1 < nsteps) step_body(step + 1, rb2, rb2);   // consumes rb2 (step+2), refills rb2 with step+4
    }

    // ---- epilogue (as conv_igemm_bf16) ----
    const int col_l = lane & 31, rsh = 4 * (lane >> 5);
    if (p.stats) {
        float* red = reinterpret_cast<float*>(As);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
                    const float v = row_out[row] >= 0 ? acc[i][j][r] : 0.f;
                    s1 += v; s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int c = wn * TN * 32 + j * 32 + lane;
                red[(wm * BN + c) * 2 + 0] = s1;
                red[(wm * BN + c) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int wmi = 0; wmi < WAVES_M; ++wmi) { s1 += red[(wmi * BN + tid) * 2]; s2 += red[(wmi * BN + tid) * 2 + 1]; }
            p.stats[((size_t)slot * 2 + 0) * p.N + n0 + tid] = s1;
            p.stats[((size_t)slot * 2 + 1) * p.N + n0 + tid] = s2;
        }
    }
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* ad = reinterpret_cast<const unsigned short*>(p.addsrc);
    const bool has_affine = p.ep_scale != nullptr;
    float es[TN], et[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + col_l;
        es[j] = (has_affine && n < p.N) ? p.ep_scale[n] : 1.f;
        et[j] = (has_affine && n < p.N) ? p.ep_shift[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + rsh;
            const int op = row_out[row];
            if (op < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + col_l;
                if (n < p.N) {
                    float v = acc[i][j][r];
                    if (has_affine) v = v * es[j] + et[j];
                    if (p.act & GDN_ACT_RELU) v = fmaxf(v, 0.f);
                    if (ad) v += bf16_to_f32(ad[(size_t)op * p.ld_add + n]);
                    if (p.act & GDN_ACT_TANH) v = tanhf(v);
                    yo[(size_t)op * p.ldy + n] = f32_to_bf16(v);
                }
            }
        }
    }
}

#include "conv_ring.h"
#include "conv_ring2.h"

// ---------------------------------------------------------------------------
// Host side: geometry -> phases / tap lists, tile selection, launch.
// ---------------------------------------------------------------------------
// Second stage of a split-K launch: y = act(sum_z part[z] + addsrc), plus the per-block BatchNorm
// partial statistics the single-stage epilogue would have produced.  64 pixels per workgroup.
#define SK_ROWS 16
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ part, int ksplit, long long npix,
                                                             int N, void* __restrict__ y, int ldy,
                                                             const void* __restrict__ addsrc, int ld_add, int act,
                                                             float* __restrict__ stats, int bf16,
                                                             const float* __restrict__ ep_scale,
                                                             const float* __restrict__ ep_shift,
                                                             const void* __restrict__ bnb_y = nullptr, int ld_bnb = 0,
                                                             const float* __restrict__ bnb_co = nullptr, int bnb_relu = 0) {
    __shared__ float sh[256 * 8];
    const int cq = N >> 2;
    const int CQ = cq < 256 ? cq : 256, PY = 256 / CQ;
    const int tx = threadIdx.x % CQ, ty = threadIdx.x / CQ;
    const long long p0 = (long long)blockIdx.x * SK_ROWS;
    const long long p1 = p0 + SK_ROWS < npix ? p0 + SK_ROWS : npix;
    const size_t zs = (size_t)npix * N;
    for (int q = tx; q < cq; q += CQ) {
        const int c = q * 4;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (ty < PY) {
            for (long long pix = p0 + ty; pix < p1; pix += PY) {
                f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)pix * N + c);
                int z = 1;
                for (; z + 8 <= ksplit; z += 8) {        // eight slabs in flight, summed in slab order (same bits as one at a time)
                    f32x4 u[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) u[e] = *reinterpret_cast<const f32x4*>(part + (z + e) * zs + (size_t)pix * N + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v += u[e];
                }
                for (; z < ksplit; ++z) v += *reinterpret_cast<const f32x4*>(part + z * zs + (size_t)pix * N + c);
                if (!bnb_y) { s1 += v; s2 += v * v; }
                if (ep_scale) v = v * *reinterpret_cast<const f32x4*>(ep_scale + c) + *reinterpret_cast<const f32x4*>(ep_shift + c);
                if (act & GDN_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (addsrc) v += ld4_any(addsrc, (size_t)pix * ld_add + c, bf16);
                if (act & GDN_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
                }
                st4_any(y, (size_t)pix * ldy + c, v, bf16);
                if (bnb_y) {
                    // v is the final gradient of z = [relu](BN(bnb_y)): the BatchNorm backward's partial sums (conv_ring.h, same arithmetic)
                    const f32x4 yv = ld4_any(bnb_y, (size_t)pix * ld_bnb + c, bf16);
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnb_co + c), sf = *reinterpret_cast<const f32x4*>(bnb_co + N + c);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(bnb_co + 2 * N + c), is = *reinterpret_cast<const f32x4*>(bnb_co + 3 * N + c);
                    // (the gradient as the consumer reads it: rounded to the tensor's dtype)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float dz = bf16 ? bf16_h_to_f32(f32_to_bf16_h(v[e])) : v[e];
                        if (bnb_relu && !(yv[e] * sc[e] + sf[e] > 0.f)) dz = 0.f;
                        s1[e] += dz; s2[e] += dz * ((yv[e] - mu[e]) * is[e]);
                    }
                }
            }
        }
        if (stats) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) { sh[threadIdx.x * 8 + e] = s1[e]; sh[threadIdx.x * 8 + 4 + e] = s2[e]; }
            __syncthreads();
            if (ty == 0) {
                f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < PY; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r1[e] += sh[(j * CQ + tx) * 8 + e]; r2[e] += sh[(j * CQ + tx) * 8 + 4 + e]; }
                }
                *reinterpret_cast<f32x4*>(stats + ((size_t)blockIdx.x * 2 + 0) * N + c) = r1;
                *reinterpret_cast<f32x4*>(stats + ((size_t)blockIdx.x * 2 + 1) * N + c) = r2;
            }
        }
    }
}

// CU count of the current device (a device attribute, read once per process: 256 on MI355X)
int gdn_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
        (void)hipGetLastError();
        return v > 0 ? v : 256;
    }();
    return n;
}

namespace {


// Buffer descriptors address 32 bits; out-of-range sentinels sit just below 4 GiB.
const uint64_t kMaxBufBytes = 0xFF000000ull;

struct TileCfg { int bm, bn; };
// cfg ids: 1: 128x128  2: 128x64  3: 64x64  4: 128x32  (5: scalar-gather 128x64)  6: 32x128  7: 64x128
//          8: row-patch 256x128  9: row-patch 256x64 (bf16, stride-1 layers with k >= 3)
//          10: LDS-DMA ring 256x64  11: LDS-DMA ring 256x128 (conv_ring.h: bf16, stride-1 layers with k in {3,5,7,9})
//          12: LDS-DMA ring 512x64  13: LDS-DMA ring 512x128 (conv_ring2.h: 32-channel slabs)
#define NUM_CFG 14
const TileCfg kCfg[NUM_CFG] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {128, 32}, {128, 64}, {32, 128}, {64, 128},
                               {256, 128}, {256, 64}, {256, 64}, {256, 128}, {512, 64}, {512, 128}};

// Row-patch kernel eligibility (geometry only, so the slot/workspace queries agree with the launch): one phase,
// stride 1, >= 3 taps per filter row with consecutive dx, 64-channel slabs, and a 256-pixel tile whose touched
// image rows + halos fit the staged patch.
bool rowpatch_ok(const IgemmParams& P) {
    if (P.nphase != 1 || P.stride != 1 || (P.Cred % 64)) return false;
    const IgemmPhase& ph = P.ph[0];
    const int ntap = ph.tap_end - ph.tap_begin;
    int kw = 1;
    while (kw < ntap && P.tdy[ph.tap_begin + kw] == P.tdy[ph.tap_begin]) ++kw;
    if (kw < 3 || kw > RP_KMAX || ntap % kw || ntap / kw > RP_KMAX) return false;
    for (int r = 0; r < ntap / kw; ++r) {
        int lo = P.tdx[ph.tap_begin + r * kw], hi = lo;
        for (int t = 0; t < kw; ++t) {
            const int i = ph.tap_begin + r * kw + t;
            if (P.tdy[i] != P.tdy[ph.tap_begin + r * kw]) return false;
            lo = P.tdx[i] < lo ? P.tdx[i] : lo; hi = P.tdx[i] > hi ? P.tdx[i] : hi;
        }
        if (hi - lo != kw - 1) return false;
    }
    const int nrows_max = (RP_BM - 1 + ph.Wo - 1) / ph.Wo + 1;
    return nrows_max <= RP_NRMAX && RP_BM + nrows_max * (kw - 1) <= RP_PMAX;
}

// LDS-DMA ring kernel eligibility (geometry only): as the row-patch kernel, with an odd window of 3..9 taps per row (the
// instantiated schedules), whole BN-channel tiles, and the touched rows' halos within the 64 spare positions of the patch.
int ring_kw(const IgemmParams& P, int bm = RG_BM) {
    if (P.nphase != 1 || P.stride != 1 || (P.Cred % 64)) return 0;
    const IgemmPhase& ph = P.ph[0];
    const int ntap = ph.tap_end - ph.tap_begin;
    int kw = 1;
    while (kw < ntap && P.tdy[ph.tap_begin + kw] == P.tdy[ph.tap_begin]) ++kw;
    if (!(kw == 3 || kw == 5 || kw == 7 || kw == 9) || ntap % kw || ntap / kw > RG_KMAX) return 0;
    // every filter row: the same dx sequence, +-1 per tap, weights consecutive (the kernel derives a tap from its row's first)
    const int dir = P.tdx[ph.tap_begin + 1] - P.tdx[ph.tap_begin];
    if (dir != 1 && dir != -1) return 0;
    for (int r = 0; r < ntap / kw; ++r)
        for (int t = 0; t < kw; ++t) {
            const int i = ph.tap_begin + r * kw + t, i0 = ph.tap_begin + r * kw;
            if (P.tdy[i] != P.tdy[i0] || P.tdx[i] != P.tdx[ph.tap_begin] + dir * t || P.twi[i] != P.twi[i0] + t) return 0;
        }
    const int nrows_max = (bm - 1 + ph.Wo - 1) / ph.Wo + 1;
    if (nrows_max > (bm == RG_BM ? RG_NRMAX : RG2_NRMAX) || bm + nrows_max * (kw - 1) > (bm == RG_BM ? RG_APOS : RG2_APOS)) return 0;
    return kw;
}
bool ring_ok(const IgemmParams& P, int bn, int bm = RG_BM) { return ring_kw(P, bm) != 0 && P.N % bn == 0 && !P.x2; }

// How the persistent workgroups of conv_ring_bf16 cover the (M-tile, N-tile) units: G workgroups (one per CU), `main` units in
// full rounds, and -- when the last round would be a small fraction of one (B = 20: 65/64 of a power of two tiles at every
// level) -- its `tail` units cut into `parts` stage ranges of `sp` stages, one per workgroup, summed by splitk_combine_kernel.
struct RingPlan { int G, grid_m, grid_n, main_units, tail_units, parts, sp, slabs, main_mtiles; int64_t tail_px; };
RingPlan ring_plan(const IgemmParams& P, int cfg) {
    RingPlan r{};
    const int bm = kCfg[cfg].bm, bn = kCfg[cfg].bn;
    const IgemmPhase& ph = P.ph[0];
    const int64_t M = (int64_t)P.B * ph.Ho * ph.Wo;
    r.grid_m = (int)cdiv64(M, bm); r.grid_n = P.N / bn;
    const int units = r.grid_m * r.grid_n, units_xcd = cdiv(r.grid_m, 8) * r.grid_n, cus_xcd = P.plan_cus / 8 > 0 ? P.plan_cus / 8 : 32;
    r.G = 8 * (units_xcd < cus_xcd ? units_xcd : cus_xcd);
    const int kw = ring_kw(P, bm), nstage = kw ? (ph.tap_end - ph.tap_begin) / kw * (P.Cred / (bm == RG_BM ? 64 : 32)) : 0;
    const int rounds = units / r.G, tail = units % r.G;
    r.main_units = units; r.parts = 1; r.sp = nstage; r.slabs = 0; r.main_mtiles = r.grid_m; r.tail_px = 0;
    if (P.ring_notail) return r;                                       // (tile_cfg & 0x800: every unit whole)
    if (rounds >= 1 && tail > 0 && 4 * tail <= r.G && nstage >= 2 && r.G % r.grid_n == 0) {
        int parts = r.G / tail < nstage ? r.G / tail : nstage;
        if (parts > 8) parts = 8;                 // (each range is one fp32 slab -- two with the 256 x 64 form -- that the combine pass reads back)
        r.sp = cdiv(nstage, parts);
        r.parts = cdiv(nstage, r.sp);
        r.main_units = rounds * r.G; r.tail_units = tail;
        r.main_mtiles = r.main_units / r.grid_n;
        r.tail_px = M - (int64_t)r.main_mtiles * bm;
        r.slabs = r.parts * (cfg == 10 ? 2 : 1);
    }
    return r;
}
size_t ring_ws_bytes(const IgemmParams& P, int cfg) {
    if (cfg < 10) return 0;
    const RingPlan r = ring_plan(P, cfg);
    return r.parts > 1 ? (size_t)r.slabs * r.tail_px * P.N * sizeof(float) : 0;
}

// bf16: the MFMA is 16x faster, so tiles must be large enough to amortise the staging of a k-step.
int pick_cfg_bf16(const IgemmParams& P, int64_t M, int N, int forced) {
    const bool rp = rowpatch_ok(P);
    if (forced >= 1 && forced <= 3) return forced;
    if ((forced == 8 || forced == 9) && rp) return forced;
    if (forced == 10 && ring_ok(P, 64)) return 10;
    if (forced == 11 && ring_ok(P, 128)) return 11;
    if (forced == 12 && ring_ok(P, 64, RG2_BM)) return 12;
    // round 4 (tests/diag/ring_check.py, ring_probe.py at B = 20): the LDS-DMA ring kernel (conv_ring.h) beats both round-1
    // kernels on every stride-1 layer with a 3..9 window -- 9x9 / 64 ch 1070 vs 930, 7x7 / 128 ch 1045 vs 886, 5x5 / 256 ch 885
    // vs 725, 3x3 / 512 ch at 8x26 418 vs 266 TFLOP/s, level with them at 16x52 (603) -- and its data gradients with a residual
    // by more (coalesced epilogue).  256 x 128 tiles where they still fill the chip's 256 persistent workgroups twice, else
    // 256 x 64 (more, smaller units for the tail: B = 20 gives 65/64 of a power of two tiles at every level).
    if (forced == 0 && ring_kw(P) != 0 && !P.x2 && N % 64 == 0) {
        // 256 x 128 tiles (0.75 fragment reads per MFMA instead of 1, no exchange of reduction halves) wherever they give every
        // CU at least one unit; else 256 x 64 (8x26 level: 68 units of 256 x 128 for 256 CUs)
        if (N % 128 == 0 && cdiv64(M, RG_BM) * (N / 128) >= P.plan_cus) return 11;
        // round 5: 512 x 64 tiles on 32-channel slabs (conv_ring2.h) for the 64-output-channel layers with wide windows -- half
        // the per-tile set-up and epilogue barriers per pixel, 2/3 of the LDS-DMA bytes (tests/diag/ring2_check.py at B = 20:
        // 9x9 / 64 ch 1104 vs 1070 TFLOP/s, reflect 7x7 128 -> 64 1101 vs 1048; N >= 128 layers lose 4-8 %)
        if (N == 64 && ring_kw(P, RG2_BM) >= 7 && cdiv64(M, RG2_BM) >= 2 * P.plan_cus) return 12;
        return 10;
    }
    if (rp && N <= 128 && cdiv64(M, RP_BM) * cdiv(N, 64) >= 448) return 9;
    if (N <= 64) return 2;
    return cdiv64(M, 128) * cdiv(N, 128) >= 384 ? 1 : 3;
}

int pick_cfg_f32(int64_t M, int N, bool scalar, int forced) {
    // Measured on MI355X at B=20 (tools/tune_conv.py, profiles/r01_tune_conv_*): 64x64 tiles at
    // 6 waves/SIMD beat every larger tile on every layer (CU balance + latency hiding).
    (void)M;
    if (scalar) return 5;
    if ((forced >= 1 && forced <= 4) || forced == 6 || forced == 7) return forced;
    if (N <= 32) return 4;
    return 3;
}

#define CFG_BF16 0x10000     // GDN_CFG_BF16 in tile_cfg: tensors hold bf16
int64_t max_phase_m(const IgemmParams& P);
// the row count the PLANS (tile configuration, split-K factor) are chosen for: the launch's own, unless GDN_PLAN_BATCH
// overrides the batch size (common.h)
int64_t plan_phase_m(const IgemmParams& P) { return max_phase_m(P) / P.B * P.plan_B; }
int pick_cfg(const IgemmParams& P, int N, bool scalar, int tile_cfg) {
    const int64_t M = plan_phase_m(P);
    return (tile_cfg & CFG_BF16) ? pick_cfg_bf16(P, M, N, tile_cfg & 0xff) : pick_cfg_f32(M, N, scalar, tile_cfg & 0xff);
}

// Builds the phase decomposition of a "transposed-type" gather:
//   out[o] += in[(o + pp - k)/s] * w[k]  for taps with (o + pp - k) % s == 0.
// Used by ConvTranspose2d forward and by Conv2d data gradients.
void build_transposed_phases(IgemmParams& P, int k, int s, int pp, int Hout, int Wout, int B) {
    P.nphase = s * s;
    P.stride = 1;
    P.osy = s; P.osx = s;
    int nt = 0;
    for (int a = 0; a < s; ++a)
        for (int b = 0; b < s; ++b) {
            IgemmPhase& ph = P.ph[a * s + b];
            ph.oy0 = a; ph.ox0 = b;
            ph.Ho = (Hout - a + s - 1) / s; ph.Wo = (Wout - b + s - 1) / s;
            ph.tap_begin = nt;
            for (int ky = 0; ky < k; ++ky) {
                if (((a + pp - ky) % s + s) % s != 0) continue;
                for (int kx = 0; kx < k; ++kx) {
                    if (((b + pp - kx) % s + s) % s != 0) continue;
                    // exact division (numerator is a multiple of s, may be negative)
                    const int ny = a + pp - ky, nx = b + pp - kx;
                    P.tdy[nt] = (short)(ny >= 0 ? ny / s : -((-ny) / s));
                    P.tdx[nt] = (short)(nx >= 0 ? nx / s : -((-nx) / s));
                    P.twi[nt] = (short)(ky * k + kx);
                    ++nt;
                }
            }
            ph.tap_end = nt;
        }
}

void build_direct_phase(IgemmParams& P, int k, int s, int pad, int Ho, int Wo) {
    P.nphase = 1;
    P.stride = s;
    P.osy = 1; P.osx = 1;
    IgemmPhase& ph = P.ph[0];
    ph.oy0 = 0; ph.ox0 = 0; ph.Ho = Ho; ph.Wo = Wo; ph.tap_begin = 0;
    int nt = 0;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            P.tdy[nt] = (short)(ky - pad); P.tdx[nt] = (short)(kx - pad); P.twi[nt] = (short)(ky * k + kx);
            ++nt;
        }
    ph.tap_end = nt;
}

int64_t max_phase_m(const IgemmParams& P) {
    int64_t m = 0;
    for (int i = 0; i < P.nphase; ++i) {
        const int64_t v = (int64_t)P.B * P.ph[i].Ho * P.ph[i].Wo;
        if (v > m) m = v;
    }
    return m;
}

template <int BM, int BN, int WM, int WN, bool SC>
void launch_one(const IgemmParams& P, hipStream_t st) {
    const int gm_pad = cdiv(P.grid_m, 8) * 8;
    dim3 grid((unsigned)(gm_pad * P.grid_n), (unsigned)P.nphase, (unsigned)P.ksplit);
    // 64-channel slabs halve the barriers and the per-tap address work per MFMA; they need the
    // reduction channels (and the concat split) to be multiples of 64.
    if constexpr (!SC && BM * BN <= 64 * 128) {
        if (P.kc == 64) {
            hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, SC, 64>), grid, dim3(256), 0, st, P);
            return;
        }
    }
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, SC, 32>), grid, dim3(256), 0, st, P);
}

// Split-K over filter taps for launches that cannot fill the chip: a level-4 3x3 layer (M = 4160) is
// 520 tiles of 64x64 for 256 CUs -- 2.03 tiles per CU, i.e. a third of the CUs wait for the ones that
// drew three.  Splitting the taps 3-4 ways gives every CU ~8 smaller units (and the level-3 layers four
// rounds instead of 1.35); the partials cost one extra pass over the (small) output.
int pick_ksplit(const IgemmParams& P, int cfg, bool scalar, int tile_cfg) {
    if (scalar || (P.N % 4) || cfg == 4 || cfg >= 8 || (tile_cfg & 0x800)) return 1;
    const TileCfg tc = kCfg[cfg];
    const int64_t blocks = cdiv64(cdiv64(plan_phase_m(P), tc.bm), 8) * 8 * cdiv(P.N, tc.bn) * P.nphase;
    if (blocks >= 1200 && !(tile_cfg & 0x400)) return 1;      // measured: splitting only pays below ~one round of resident workgroups (0x400: tuning override)
    int min_taps = MAX_TAPS;
    for (int i = 0; i < P.nphase; ++i) {
        const int n = P.ph[i].tap_end - P.ph[i].tap_begin;
        if (n < min_taps) min_taps = n;
    }
    // fp32: ~3 rounds of resident workgroups, at most 4 ways.  bf16: the fp32 partials cost relatively 16x more against
    // the faster MFMA, so split less (measured, profiles/r01_tune_conv_bf16_rowpatch.txt): fill ~one round, at most 3 ways.
    const bool bf = (tile_cfg & CFG_BF16) != 0;
    int ks = (int)cdiv64(bf ? 1024 : 4608, blocks);
    const int ks_max = (tile_cfg >> 12) & 15 ? (tile_cfg >> 12) & 15 : (bf ? 3 : 4);     // bits 12..15: tuning override
    if (ks > ks_max) ks = ks_max;
    if (ks > min_taps) ks = min_taps;
    if (ks < 1) ks = 1;
    // the slowest split sets the pace: 9 taps four ways is 2,2,2,3 -- no faster than 3,3,3 but a quarter more partials
    while (ks > 1 && cdiv(min_taps, ks - 1) == cdiv(min_taps, ks)) --ks;
    return ks;
}

size_t ksplit_bytes(const IgemmParams& P, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * P.B * P.Hy * P.Wy * P.N * sizeof(float) : 0;
}

// Launches the kernel; with ksplit > 1 the partials go to `split_ws` and the combine kernel writes
// y / stats (P.y, P.ldy, P.addsrc, P.act, P.stats keep their single-stage meaning).
int launch_igemm(IgemmParams& P, int cfg, hipStream_t st, int ksplit = 1, void* split_ws = nullptr) {
    const TileCfg tc = kCfg[cfg];
    P.grid_m = (int)cdiv64(max_phase_m(P), tc.bm);
    P.grid_n = cdiv(P.N, tc.bn);
    P.ksplit = ksplit;
    P.part = (float*)split_ws;
    P.npix_out = (long long)P.B * P.Hy * P.Wy;
    if (ksplit > 1 && !split_ws) return GDN_ERR_WORKSPACE;
    if (P.bf16) {
        const int gm_pad = cdiv(P.grid_m, 8) * 8;
        dim3 grid((unsigned)(gm_pad * P.grid_n), (unsigned)P.nphase, (unsigned)P.ksplit);
        switch (cfg) {
            case 1: hipLaunchKernelGGL((conv_igemm_bf16<128, 128, 2, 2>), grid, dim3(256), 0, st, P); break;
            case 2: hipLaunchKernelGGL((conv_igemm_bf16<128, 64, 2, 2>), grid, dim3(256), 0, st, P); break;
            case 3: hipLaunchKernelGGL((conv_igemm_bf16<64, 64, 2, 2>), grid, dim3(256), 0, st, P); break;
            case 8: hipLaunchKernelGGL((conv_rowpatch_bf16<128, 2, 2>), dim3(gm_pad * P.grid_n), dim3(256), 0, st, P); break;
            case 9: hipLaunchKernelGGL((conv_rowpatch_bf16<64, 4, 1>), dim3(gm_pad * P.grid_n), dim3(256), 0, st, P); break;
            case 10: case 11: case 12: case 13: {
                // persistent workgroups, one per CU: every XCD (workgroups b, b + 8, ...) walks its band of tiles
                const RingPlan rp = ring_plan(P, cfg);
                const dim3 g1(rp.G), b1(512);
                P.ksplit = rp.parts; P.ring_main = rp.main_units; P.ring_sp = rp.sp;
                if (rp.parts > 1 && !split_ws) return GDN_ERR_WORKSPACE;
#define GDN_RING(BNV, KWV) do { if (P.bnb_y) hipLaunchKernelGGL((conv_ring_bf16<BNV, KWV, 0, true>), g1, b1, 0, st, P); \
                                else hipLaunchKernelGGL((conv_ring_bf16<BNV, KWV>), g1, b1, 0, st, P); } while (0)
#define GDN_RING2(BNV, KWV) do { if (P.bnb_y) hipLaunchKernelGGL((conv_ring2_bf16<BNV, KWV, 0, true>), g1, b1, 0, st, P); \
                                 else hipLaunchKernelGGL((conv_ring2_bf16<BNV, KWV>), g1, b1, 0, st, P); } while (0)
                const int kw = ring_kw(P, tc.bm);
                if (cfg == 12) { if (kw == 3) GDN_RING2(64, 3); else if (kw == 5) GDN_RING2(64, 5); else if (kw == 7) GDN_RING2(64, 7); else if (kw == 9) GDN_RING2(64, 9); else return GDN_ERR_UNSUPPORTED; }
                else if (cfg == 13) return GDN_ERR_UNSUPPORTED;    // (512 x 128: 128 accumulator registers per lane of 256 -- spills)
                else if (cfg == 10) { if (kw == 3) GDN_RING(64, 3); else if (kw == 5) GDN_RING(64, 5); else if (kw == 7) GDN_RING(64, 7); else if (kw == 9) GDN_RING(64, 9); else return GDN_ERR_UNSUPPORTED; }
                else { if (kw == 3) GDN_RING(128, 3); else if (kw == 5) GDN_RING(128, 5); else if (kw == 7) GDN_RING(128, 7); else if (kw == 9) GDN_RING(128, 9); else return GDN_ERR_UNSUPPORTED; }
#undef GDN_RING
#undef GDN_RING2
                if (rp.parts > 1) {
                    // the tail ranges' slabs -> y / stats for the pixels [main_mtiles * 256, M) (a dense pixel range: ring layers have
                    // one phase and unit stride, so output pixel index == m)
                    const int64_t p0 = (int64_t)rp.main_mtiles * tc.bm;
                    const int blocks = (int)cdiv64(rp.tail_px, SK_ROWS);
                    hipLaunchKernelGGL(splitk_combine_kernel, dim3(blocks), dim3(256), 0, st, (const float*)P.part, rp.slabs, (long long)rp.tail_px,
                                       P.N, (void*)((unsigned short*)P.y + p0 * P.ldy), P.ldy,
                                       P.addsrc ? (const void*)((const unsigned short*)P.addsrc + p0 * P.ld_add) : (const void*)nullptr, P.ld_add,
                                       P.act, P.stats ? P.stats + (size_t)rp.main_mtiles * 2 * P.N : (float*)nullptr, 1, P.ep_scale, P.ep_shift,
                                       P.bnb_y ? (const void*)((const unsigned short*)P.bnb_y + p0 * P.ld_bnb) : (const void*)nullptr, P.ld_bnb,
                                       P.bnb_co, P.bnb_relu);
                }
                return gdn_launch_status();
            }
            default: return GDN_ERR_BAD_ARG;
        }
    } else
    switch (cfg) {
        case 1: launch_one<128, 128, 2, 2, false>(P, st); break;
        case 2: launch_one<128, 64, 2, 2, false>(P, st); break;
        case 3: launch_one<64, 64, 2, 2, false>(P, st); break;
        case 4: launch_one<128, 32, 4, 1, false>(P, st); break;
        case 5: launch_one<128, 64, 2, 2, true>(P, st); break;
        case 6: launch_one<32, 128, 1, 4, false>(P, st); break;
        case 7: launch_one<64, 128, 2, 2, false>(P, st); break;
        default: return GDN_ERR_BAD_ARG;
    }
    if (ksplit > 1) {
        const int blocks = (int)cdiv64(P.npix_out, SK_ROWS);
        hipLaunchKernelGGL(splitk_combine_kernel, dim3(blocks), dim3(256), 0, st, (const float*)P.part, ksplit,
                           P.npix_out, P.N, (void*)P.y, P.ldy, (const void*)P.addsrc, P.ld_add, P.act, P.stats, P.bf16,
                           P.ep_scale, P.ep_shift);
    }
    return gdn_launch_status();
}

bool geom_ok(const gdn_conv_geom* g) {
    if (!g || g->B <= 0 || g->H <= 0 || g->W <= 0 || g->Cin <= 0 || g->Cout <= 0) return false;
    if (g->k < 1 || g->k * g->k > MAX_TAPS || g->stride < 1 || g->stride > 2 || g->pad < 0) return false;
    if (g->transposed && g->pad_mode != 0) return false;
    if (g->pad_mode == 1 && (g->pad >= g->H || g->pad >= g->W)) return false;
    return true;
}

__global__ void reflect_fold_kernel(const void* __restrict__ dxp, void* __restrict__ dx,
                                    const void* __restrict__ addsrc, int ld_add, int ldx,
                                    int B, int H, int W, int C, int p, int bf16) {
    // dx[y][x] = sum of dxp over the padded coordinates that reflect onto (y, x).
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const int c4n = C / 4;
    const int64_t total = (int64_t)B * H * W * c4n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int qy[3], qx[3], ny = 0, nx = 0;
        qy[ny++] = y + p;
        if (y >= 1 && y <= p) qy[ny++] = p - y;
        if (y <= H - 2 && y >= H - 1 - p) qy[ny++] = 2 * (H - 1) - y + p;
        qx[nx++] = x + p;
        if (x >= 1 && x <= p) qx[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) qx[nx++] = 2 * (W - 1) - x + p;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e)
                s += ld4_any(dxp, ((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c4 * 4, bf16);
        const size_t op = (size_t)(b * H + y) * W + x;
        if (addsrc) s += ld4_any(addsrc, op * ld_add + c4 * 4, bf16);
        st4_any(dx, op * ldx + c4 * 4, s, bf16);
    }
}

// Same fold for channel counts that are not a multiple of 4 (the 1- and 3-channel network inputs), fp32 only.
__global__ void reflect_fold_scalar_kernel(const float* __restrict__ dxp, float* __restrict__ dx,
                                           const float* __restrict__ addsrc, int ld_add, int ldx,
                                           int B, int H, int W, int C, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int qy[3], qx[3], ny = 0, nx = 0;
        qy[ny++] = y + p;
        if (y >= 1 && y <= p) qy[ny++] = p - y;
        if (y <= H - 2 && y >= H - 1 - p) qy[ny++] = 2 * (H - 1) - y + p;
        qx[nx++] = x + p;
        if (x >= 1 && x <= p) qx[nx++] = p - x;
        if (x <= W - 2 && x >= W - 1 - p) qx[nx++] = 2 * (W - 1) - x + p;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e) s += dxp[((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c];
        const size_t op = (size_t)(b * H + y) * W + x;
        if (addsrc) s += addsrc[op * ld_add + c];
        dx[op * ldx + c] = s;
    }
}

}  // namespace

extern "C" int gdn_conv_out_dims(const gdn_conv_geom* g, int32_t* Ho, int32_t* Wo) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!geom_ok(g)) return GDN_ERR_BAD_ARG;
    if (g->transposed) {
        *Ho = (g->H - 1) * g->stride - 2 * g->pad + g->k;
        *Wo = (g->W - 1) * g->stride - 2 * g->pad + g->k;
    } else {
        *Ho = (g->H + 2 * g->pad - g->k) / g->stride + 1;
        *Wo = (g->W + 2 * g->pad - g->k) / g->stride + 1;
    }
    return (*Ho > 0 && *Wo > 0) ? GDN_OK : GDN_ERR_BAD_ARG;
}

// conv_head.hip: register-blocked VALU kernel for the 1-channel 9x9 heads
int gdn_conv_head_fwd(const gdn_conv_geom* g, const void* x, int32_t ldx, const float* w, float* y, int32_t ldy,
                      int32_t act, int32_t x_bf16, void* stream);

static int fill_fwd(const gdn_conv_geom* g, IgemmParams& P) {
    int Ho, Wo;
    if (gdn_conv_out_dims(g, &Ho, &Wo) != GDN_OK) return GDN_ERR_BAD_ARG;
    P.B = g->B; P.Hi = g->H; P.Wi = g->W; P.N = g->Cout; P.Cred = g->Cin;
    P.plan_B = gdn_plan_batch(g); P.plan_cus = gdn_plan_cus(g);
    P.w_tap_stride = g->Cout * g->Cin;
    P.Hy = Ho; P.Wy = Wo;
    P.pad_mode = g->pad_mode;
    if (g->transposed) build_transposed_phases(P, g->k, g->stride, g->pad, Ho, Wo, g->B);
    else build_direct_phase(P, g->k, g->stride, g->pad, Ho, Wo);
    return GDN_OK;
}

// Slot count written for the configuration the launcher picks.
extern "C" int64_t gdn_conv_stats_slots(const gdn_conv_geom* g, int32_t tile_cfg) {
    IgemmParams P{};
    if (fill_fwd(g, P) != GDN_OK) return GDN_ERR_BAD_ARG;
    const bool scalar = (g->Cin % KC_MIN) != 0;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, g->Cout, scalar, tile_cfg);
    if (pick_ksplit(P, cfg, scalar, tile_cfg) > 1) return cdiv64((int64_t)P.B * P.Hy * P.Wy, SK_ROWS);
    if (cfg >= 10) {
        const RingPlan rp = ring_plan(P, cfg);
        if (rp.parts > 1) return rp.main_mtiles + cdiv64(rp.tail_px, SK_ROWS);
    }
    return (int64_t)P.nphase * cdiv64(max_phase_m(P), kCfg[cfg].bm);
}

extern "C" size_t gdn_conv_fwd_workspace_bytes(const gdn_conv_geom* g, int32_t tile_cfg) {
    IgemmParams P{};
    if (fill_fwd(g, P) != GDN_OK) return 0;
    const bool scalar = (g->Cin % KC_MIN) != 0;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, g->Cout, scalar, tile_cfg);
    return ksplit_bytes(P, pick_ksplit(P, cfg, scalar, tile_cfg)) + ring_ws_bytes(P, cfg);
}

extern "C" int gdn_conv_fwd(const gdn_conv_geom* g, const void* xv, int32_t ldx, const void* x2v, int32_t ldx2,
                            int32_t C1, const void* wv, void* yv, int32_t ldy, const void* addsrcv, int32_t ld_add,
                            float* stats, const float* ep_scale, const float* ep_shift, int32_t act, int32_t tile_cfg,
                            void* workspace, size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    const float *x = (const float*)xv, *x2 = (const float*)x2v, *w = (const float*)wv, *addsrc = (const float*)addsrcv;
    float* y = (float*)yv;
    const bool bf = (tile_cfg & CFG_BF16) != 0;
    const uint64_t es = bf ? 2 : 4;
    if (!geom_ok(g) || !x || !w || !y) return GDN_ERR_BAD_ARG;
    if ((ep_scale == nullptr) != (ep_shift == nullptr)) return GDN_ERR_BAD_ARG;
    if (g->Cout == 1 && !x2 && !stats && !addsrc && !ep_scale && !(act & GDN_ACT_RELU) && (tile_cfg & ~CFG_BF16) == 0) {
        // 1-channel heads: with GDN_CFG_BF16 only x is bf16 -- weights and the depth map stay fp32
        const int rc = gdn_conv_head_fwd(g, x, ldx, w, y, ldy, act, bf ? 1 : 0, stream);
        if (rc != GDN_ERR_UNSUPPORTED || bf) return rc;
    }
    IgemmParams P{};
    if (fill_fwd(g, P) != GDN_OK) return GDN_ERR_BAD_ARG;
    const bool scalar = (g->Cin % KC_MIN) != 0;
    if (x2 == nullptr) C1 = g->Cin;
    if (C1 <= 0 || C1 > g->Cin) return GDN_ERR_BAD_ARG;
    if (scalar && C1 != g->Cin) return GDN_ERR_UNSUPPORTED;
    if (!scalar && ((C1 % KC_MIN) || (ldx % 4) || (x2 && (ldx2 % 4)))) return GDN_ERR_UNSUPPORTED;
    // bf16: 64-channel (128-byte) slabs, 16-byte aligned rows
    if (bf && ((g->Cin % 64) || (C1 % 64) || (ldx % 8) || (x2 && (ldx2 % 8)))) return GDN_ERR_UNSUPPORTED;
    P.bf16 = bf ? 1 : 0;
    P.x = x; P.x2 = x2; P.w = w; P.y = y; P.addsrc = addsrc; P.stats = stats;
    P.ep_scale = ep_scale; P.ep_shift = ep_shift;
    P.C1 = C1; P.C2 = g->Cin - C1; P.ldx1 = ldx; P.ldx2 = ldx2; P.ldy = ldy; P.ld_add = ld_add; P.act = act;
    {
        const uint64_t npix = (uint64_t)g->B * g->H * g->W;
        const uint64_t xb = ((npix - 1) * (uint64_t)ldx + C1) * es, x2b = x2 ? ((npix - 1) * (uint64_t)ldx2 + P.C2) * es : 0;
        const uint64_t wb = (uint64_t)g->k * g->k * g->Cout * g->Cin * es;
        // per-workgroup descriptors span at most two images
        const uint64_t two1 = 2ull * g->H * g->W * (uint64_t)ldx * es, two2 = x2 ? 2ull * g->H * g->W * (uint64_t)ldx2 * es : 0;
        if (two1 >= kMaxBufBytes || two2 >= kMaxBufBytes || wb >= kMaxBufBytes) return GDN_ERR_UNSUPPORTED;
        P.x_bytes = xb; P.x2_bytes = x2b; P.w_bytes = (unsigned)wb;
    }
    // 32-channel slabs measured faster than 64 everywhere (6 waves/SIMD vs 4); 0x200 selects 64 for tuning runs
    P.kc = (!scalar && g->Cin % 64 == 0 && C1 % 64 == 0 && (tile_cfg & 0x200)) ? 64 : 32;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, P.N, scalar, tile_cfg);
    // the row-patch kernel reads one input tensor; a fused concat with k >= 3 has no call site in the networks
    if (cfg >= 8 && x2) return GDN_ERR_UNSUPPORTED;
    if (cfg >= 10) P.kc = (tile_cfg >> 12) & 15;             // conv_ring_bf16: measurement knobs (0 in production)
    // conv_ring_bf16's epilogue moves 16 bytes = 8 channels per lane on y and addsrc (the header's "pixel pitches
    // multiples of 8" enforced on the output side too)
    if (cfg >= 10 && ((ldy % 8) || (addsrc && (ld_add % 8)))) return GDN_ERR_UNSUPPORTED;
    const int ksplit = pick_ksplit(P, cfg, scalar, tile_cfg);
    if (ksplit > 1 && (!workspace || workspace_bytes < ksplit_bytes(P, ksplit))) return GDN_ERR_WORKSPACE;
    if (ring_ws_bytes(P, cfg) && (!workspace || workspace_bytes < ring_ws_bytes(P, cfg))) return GDN_ERR_WORKSPACE;
    if ((ksplit > 1 || ring_ws_bytes(P, cfg)) && ((ldy % 4) || (addsrc && (ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    return launch_igemm(P, cfg, (hipStream_t)stream, ksplit, workspace);
}

// Geometry of a data-gradient launch (everything but the pointers).
static bool fill_dgrad(const gdn_conv_geom* g, IgemmParams& P, bool& fold, bool& scalar) {
    int Ho, Wo;
    if (!geom_ok(g) || gdn_conv_out_dims(g, &Ho, &Wo) != GDN_OK) return false;
    P.B = g->B; P.Hi = Ho; P.Wi = Wo;             // the gathered tensor is dy
    P.plan_B = gdn_plan_batch(g); P.plan_cus = gdn_plan_cus(g);
    P.N = g->Cin; P.Cred = g->Cout; P.C1 = g->Cout; P.C2 = 0;
    P.w_tap_stride = g->Cin * g->Cout;
    P.pad_mode = 0; P.act = GDN_ACT_NONE;
    scalar = (g->Cout % KC_MIN) != 0;
    fold = !g->transposed && g->pad_mode == 1 && g->pad > 0;
    if (g->transposed) {
        // dx_T[i] = sum_k dy_T[i*s - p + k] w[k]: a plain strided gather over dy.
        P.Hy = g->H; P.Wy = g->W;
        build_direct_phase(P, g->k, g->stride, g->pad, g->H, g->W);
    } else if (!fold) {
        P.Hy = g->H; P.Wy = g->W;
        build_transposed_phases(P, g->k, g->stride, g->pad, g->H, g->W, g->B);
    } else {
        P.Hy = g->H + 2 * g->pad; P.Wy = g->W + 2 * g->pad;
        build_transposed_phases(P, g->k, g->stride, 0, P.Hy, P.Wy, g->B);
    }
    return true;
}

static size_t fold_bytes(const gdn_conv_geom* g) {
    const size_t b = (size_t)g->B * (g->H + 2 * g->pad) * (g->W + 2 * g->pad) * g->Cin * sizeof(float);
    return (b + 255) / 256 * 256;
}

extern "C" size_t gdn_conv_dgrad_workspace_bytes(const gdn_conv_geom* g, int32_t tile_cfg) {
    IgemmParams P{};
    bool fold, scalar;
    if (!fill_dgrad(g, P, fold, scalar)) return 0;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, P.N, scalar, tile_cfg);
    return (fold ? fold_bytes(g) : 0) + ksplit_bytes(P, pick_ksplit(P, cfg, scalar, tile_cfg)) + ring_ws_bytes(P, cfg);
}

// slots of the BatchNorm-backward partials the data gradient's epilogue can emit for this layer (0: not available -- only the
// LDS-DMA ring kernel of the bf16 stride-1 layers without reflection padding does it)
extern "C" int64_t gdn_conv_dgrad_bnb_slots(const gdn_conv_geom* g, int32_t tile_cfg) {
    IgemmParams P{};
    bool fold, scalar;
    if (!g || !fill_dgrad(g, P, fold, scalar) || fold || scalar || !(tile_cfg & CFG_BF16)) return 0;
    P.bf16 = 1;
    if ((g->Cout % 64) || (g->Cin % 8)) return 0;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, P.N, scalar, tile_cfg);
    if (cfg < 10) return 0;
    const RingPlan rp = ring_plan(P, cfg);
    if (rp.parts > 1) return rp.main_mtiles + cdiv64(rp.tail_px, SK_ROWS);
    return (int64_t)P.nphase * cdiv64(max_phase_m(P), kCfg[cfg].bm);
}

extern "C" int gdn_conv_dgrad(const gdn_conv_geom* g, const void* dyv, int32_t ldy, const void* wtv, void* dxv,
                              int32_t ldx, const void* addsrcv, int32_t ld_add, const void* bnb_y, int32_t ld_bnb,
                              const float* bnb_co, int32_t bnb_relu, float* bnb_partial, int32_t dx_up2x, void* workspace,
                              size_t workspace_bytes, int32_t tile_cfg, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    const float *dy = (const float*)dyv, *wt = (const float*)wtv, *addsrc = (const float*)addsrcv;
    float* dx = (float*)dxv;
    const bool bf = (tile_cfg & CFG_BF16) != 0;
    const uint64_t es = bf ? 2 : 4;
    if (!dy || !wt || !dx) return GDN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    IgemmParams P{};
    bool fold, scalar;
    if (!fill_dgrad(g, P, fold, scalar)) return GDN_ERR_BAD_ARG;
    P.x = dy; P.ldx1 = ldy; P.w = wt;
    {
        const uint64_t xb = (((uint64_t)g->B * P.Hi * P.Wi - 1) * (uint64_t)ldy + g->Cout) * es;
        const uint64_t wb = (uint64_t)g->k * g->k * g->Cout * g->Cin * es;
        if (2ull * P.Hi * P.Wi * (uint64_t)ldy * es >= kMaxBufBytes || wb >= kMaxBufBytes) return GDN_ERR_UNSUPPORTED;
        P.x_bytes = xb; P.x2_bytes = 0; P.w_bytes = (unsigned)wb;
    }
    if (!scalar && (ldy % 4)) return GDN_ERR_UNSUPPORTED;
    if (bf && (scalar || (g->Cout % 64) || (ldy % 8) || (g->Cin % 4))) return GDN_ERR_UNSUPPORTED;
    P.bf16 = bf ? 1 : 0;
    if (fold && (g->Cin % 4) && bf) return GDN_ERR_UNSUPPORTED;
    // dx_up2x (1: align_corners False, 2: True): the layer's input was the x2 bilinear upsampling of a tensor with no other
    // consumer of the upsampled form; dx / addsrc are THAT tensor's gradient [B, H/2, W/2, Cin] -- the fold pass of a reflection
    // layer applies the adjoint of the interpolation as it folds (up2x.h), the full-resolution gradient is never written
    if (dx_up2x && (dx_up2x < 0 || dx_up2x > 2 || !fold || (g->Cin % 4) || (g->H & 1) || (g->W & 1) || bnb_y)) return GDN_ERR_UNSUPPORTED;
    P.ring_notail = (tile_cfg & 0x800) ? 1 : 0;
    const int cfg = pick_cfg(P, P.N, scalar, tile_cfg);
    const int ksplit = pick_ksplit(P, cfg, scalar, tile_cfg);
    const size_t rb = ring_ws_bytes(P, cfg);
    const size_t fb = fold ? fold_bytes(g) : 0, need = fb + ksplit_bytes(P, ksplit) + rb;
    if (need && (!workspace || workspace_bytes < need)) return GDN_ERR_WORKSPACE;
    if (fold) { P.y = (float*)workspace; P.ldy = g->Cin; P.addsrc = nullptr; P.ld_add = 0; }
    else { P.y = dx; P.ldy = ldx; P.addsrc = addsrc; P.ld_add = ld_add; }
    if ((ksplit > 1 || rb) && ((P.ldy % 4) || (P.addsrc && (P.ld_add % 4)))) return GDN_ERR_UNSUPPORTED;
    if (cfg >= 10 && ((P.ldy % 8) || (P.addsrc && (P.ld_add % 8)) || (bnb_y && (ld_bnb % 8)))) return GDN_ERR_UNSUPPORTED;   // (16-byte epilogue accesses)
    P.kc = (!scalar && g->Cout % 64 == 0 && (tile_cfg & 0x200)) ? 64 : 32;
    if (cfg >= 10) P.kc = (tile_cfg >> 12) & 15;
    if (bnb_y) {
        // the producer BatchNorm's backward partials from this epilogue: gdn_conv_dgrad_bnb_slots(g, tile_cfg) said so
        if (!bnb_co || !bnb_partial || (ld_bnb % 8)) return GDN_ERR_BAD_ARG;
        if (cfg < 10 || fold || !bf) return GDN_ERR_UNSUPPORTED;
        P.bnb_y = bnb_y; P.ld_bnb = ld_bnb; P.bnb_co = bnb_co; P.bnb_relu = bnb_relu ? 1 : 0; P.stats = bnb_partial;
    }
    int rc = launch_igemm(P, cfg, st, ksplit, (ksplit > 1 || rb) ? (char*)workspace + fb : nullptr);
    if (rc != GDN_OK) return rc;
    if (fold && (g->Cin % 4)) {
        const int64_t total = (int64_t)g->B * g->H * g->W * g->Cin;
        const int blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
        hipLaunchKernelGGL(reflect_fold_scalar_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dx, addsrc,
                           ld_add, ldx, g->B, g->H, g->W, g->Cin, g->pad);
        rc = gdn_launch_status();
    } else if (fold && dx_up2x) {
        const bool w8 = (g->Cin % 8) == 0 && (ldx % 8) == 0 && (!addsrc || (ld_add % 8) == 0);
        const int64_t nb = cdiv64((int64_t)g->B * (g->H / 2) * (g->W / 2) * (g->Cin / (w8 ? 8 : 4)), 256);
        hipLaunchKernelGGL(w8 ? reflect_fold_up2x8_kernel : reflect_fold_up2x_kernel, dim3((unsigned)(nb < 65536 * 8 ? nb : 65536 * 8)),
                           dim3(256), 0, st, (const void*)workspace, (void*)dx, ldx, (const void*)addsrc, ld_add, g->B, g->H, g->W,
                           g->Cin, g->pad, dx_up2x - 1, P.bf16);
        rc = gdn_launch_status();
    } else if (fold) {
        const int64_t total = (int64_t)g->B * g->H * g->W * (g->Cin / 4);
        const int blocks = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
        hipLaunchKernelGGL(reflect_fold_kernel, dim3(blocks), dim3(256), 0, st, (const void*)workspace, (void*)dx,
                           (const void*)addsrc, ld_add, ldx, g->B, g->H, g->W, g->Cin, g->pad, P.bf16);
        rc = gdn_launch_status();
    }
    return rc;
}
