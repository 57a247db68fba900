// bf16 weight-gradient kernel for gfx950 (BASELINE configs[2]): bf16 activations / output gradients,
// v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 dW.
//
//   dW[ky][kx][r][c] = sum_{b,oy,ox} G[b][oy][ox][r] * X[b][oy*s - p + ky][ox*s - p + kx][c]
//
// The reduction runs over PIXELS, and both operands are NHWC (channels contiguous, pixels strided),
// so the MFMA wants both of them transposed: lane (row = channel) needs 8 consecutive k = pixels.
// gfx950's ds_read_b64_tr_b16 does that transpose inside the LDS read: tiles are staged exactly as
// they arrive from HBM ([pixel][channel] rows) and read column-major.
//
//   * a workgroup owns one filter ROW ky, 64 channels of G, 64 channels of X and a range of image
//     rows (split-K over pixels), like the fp32 kernel (conv_wgrad.hip);
//   * a SEGMENT is nr whole image rows (tw = W) or a tw-pixel piece of one row, flattened to
//     <= 208 pixels = 13 MFMA k-steps of 16 pixels -- 208 = 8x26 = 4x52 = 2x104 = 1x208 tiles every
//     level of the 128x416 pyramid without a ragged tail.  Pixels past the segment pair a zero G
//     with a finite X;
//   * LDS images are split by 32-channel half (one per wave row / wave column): rows of 64 bytes,
//     so the four pixel rows a 16-lane group transposes fall on four different 16-bank quarters
//     (conflict-free without padding or swizzle); stride-2 layers keep even and odd X positions in
//     two planes so that tap kx reads consecutive rows there too;
//   * the X position of pixel j comes from a 208-entry LDS table (row wrap + stride folded in), the
//     tap adds a compile-time multiple of the row pitch;
//   * partial slabs per split, fixed-order second-stage sum (same as fp32): bitwise reproducible.
#include "common.h"

#define WB_SEGMAX 208
#define WB_KMAX 9

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

struct WgradBfParams {
    const void* g; const void* x; float* part;
    int B, Hg, Wg, ldg, Cg;
    int Hx, Wx, ldx, Cx;
    int k, stride, pad, pad_mode;
    int n_cgt, n_cxt, S, rows_per_split;
    int tw, nr, xl, nks;            // segment: nr rows x tw pixels, xl = 256-chunk loads per X row, nks k-steps
    int npos, rp, pl;               // X positions per row; LDS rows per image row; plane size (stride 2)
    unsigned g_bytes, x_bytes;
};

__device__ __forceinline__ s16x4 tr_read(const lds_u8* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

#include "wgrad_ring.h"

template <int NTW, int XV, int GV>
__device__ __forceinline__ void wgrad_bf16_body(const WgradBfParams& p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_u8* smem = (lds_u8*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1, h = lane >> 5;
    // Workgroups of one split read the same rows of G and X: keep them on one XCD (block ids go round-robin
    // over the 8 XCDs) so the re-reads by the other filter rows / channel tiles hit that XCD's L2.
    const int base = p.k * p.n_cgt * p.n_cxt;
    const int local = blockIdx.x >> 3;
    const int split = (local / base) * 8 + (blockIdx.x & 7);
    if (split >= p.S) return;
    int id = local % base;
    const int ky = id % p.k; id /= p.k;
    const int cgt = id % p.n_cgt;
    const int cxt = id / p.n_cgt;
    const int cg0 = cgt * 64, cx0 = cxt * 64;
    const int tw = p.tw, nr = p.nr, xl = p.xl, nks = p.nks, npos = p.npos, RP = p.rp, PL = p.pl, s = p.stride;
    const int segpix = nks * 16;
    const unsigned GH = (unsigned)segpix * 64u, XH = (unsigned)nr * RP * 64u;     // bytes per 32-channel half image
    lds_u8* Gs = smem;
    lds_u8* Xs = smem + 2 * GH;
    __attribute__((address_space(3))) int* ptab = (__attribute__((address_space(3))) int*)(Xs + 2 * XH);
    for (int j = tid; j < segpix; j += 256) {
        const int rr = j / tw, xx = j - rr * tw;
        ptab[j] = j < nr * tw ? (rr * RP + xx) * 64 : 0;
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int R = p.B * p.Hg;
    const int r0 = split * p.rows_per_split;
    const int r1 = min(R, r0 + p.rows_per_split);
    const int nseg = nr > 1 ? 1 : (p.Wg + tw - 1) / tw;           // pieces per row (nr == 1)
    const int total = nr > 1 ? (r1 - r0 + nr - 1) / nr : (r1 - r0) * nseg;
    const bool ox_varies = nseg > 1;

    const unsigned OOB = 0xFFFFFF00u;
    __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0, (int)p.g_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const int pj = tid >> 3, ch = tid & 7;                         // staging: pixel/position within a 32-row pass, 16-B chunk
    const unsigned g_lane = (unsigned)(pj * p.ldg + cg0 + ch * 8) * 2u;
    const unsigned g_st = (unsigned)(ch >> 2) * GH + (unsigned)pj * 64u + (unsigned)(ch & 3) * 16u;
    const int g_pass_soff = 32 * p.ldg * 2;
    // X slot e = rr * xl + m: image row rr, pass m of that row
    unsigned x_lane[XV], x_st[XV];
    auto x_lane_of = [&](int m, int ox0) -> unsigned {
        const int pos = pj + 32 * m;
        int ix = ox0 * s - p.pad + pos;
        if (p.pad_mode == 1) ix = reflect_idx(ix, p.Wx);
        const bool ok = pos < npos && ix >= 0 && ix < p.Wx;
        return ok ? (unsigned)(ix * p.ldx + cx0 + ch * 8) * 2u : OOB;
    };
#pragma unroll
    for (int e = 0; e < XV; ++e) {
        const int rr = e / xl, m = e - rr * xl;
        const int pos = pj + 32 * m;
        x_lane[e] = rr < nr ? x_lane_of(m, 0) : OOB;
        const int row = rr * RP + (s == 2 ? (pos & 1) * PL + (pos >> 1) : pos);
        x_st[e] = (rr < nr && pos < npos) ? (unsigned)(ch >> 2) * XH + (unsigned)row * 64u + (unsigned)(ch & 3) * 16u
                                          : 0xFFFFFFFFu;
    }

    f32x4 rg[GV], rx[XV];
    auto gload = [&](int sidx) {
        int rseg, ox0, nvalid;
        if (nr > 1) { rseg = r0 + sidx * nr; ox0 = 0; nvalid = min(nr, r1 - rseg) * tw; }
        else { rseg = r0 + sidx / nseg; ox0 = (sidx % nseg) * tw; nvalid = min(tw, p.Wg - ox0); }
        const int soff_g = (int)((unsigned)((rseg * p.Wg + ox0) * p.ldg) * 2u);
#pragma unroll
        for (int e = 0; e < GV; ++e)
            rg[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs_g, pj + 32 * e < nvalid ? g_lane : OOB, soff_g + e * g_pass_soff, 0));
        if (ox_varies) {
#pragma unroll
            for (int e = 0; e < XV; ++e) x_lane[e] = e < xl ? x_lane_of(e, ox0) : OOB;
        }
        int b = rseg / p.Hg, oy = rseg - b * p.Hg;
        int rr_prev = -1, soff_x = 0;
        bool rok = false;
#pragma unroll
        for (int e = 0; e < XV; ++e) {
            const int rr = e / xl;
            if (rr != rr_prev) {                       // wave-uniform: next image row of the segment
                if (rr_prev >= 0 && ++oy == p.Hg) { oy = 0; ++b; }
                rr_prev = rr;
                int iy = oy * s - p.pad + ky;
                if (p.pad_mode == 1) iy = reflect_idx(iy, p.Hx);
                rok = rr < nr && rseg + rr < r1 && iy >= 0 && iy < p.Hx;
                soff_x = rok ? (int)((unsigned)((b * p.Hx + iy) * p.Wx * p.ldx) * 2u) : 0;
            }
            rx[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, rok ? x_lane[e] : OOB, soff_x, 0));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int e = 0; e < GV; ++e)
            if (pj + 32 * e < segpix) *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(Gs + g_st + e * 2048) = rg[e];
#pragma unroll
        for (int e = 0; e < XV; ++e)
            if (x_st[e] != 0xFFFFFFFFu) *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(Xs + x_st[e]) = rx[e];
    };

    // fragment addressing: lane = 32h + 16g + 4q + pp supplies row (pixel) 8h + q, bytes 32g + 8pp of the 64-B row
    const int q = (lane >> 2) & 3;
    const unsigned frag = (unsigned)((lane >> 4) & 1) * 32u + (unsigned)(lane & 3) * 8u;
    const lds_u8* Ga = Gs + (unsigned)wr * GH + (unsigned)(8 * h + q) * 64u + frag;
    const lds_u8* Xb = Xs + (unsigned)wc * XH + frag;
    const __attribute__((address_space(3))) int* pt = ptab + 8 * h + q;
    unsigned tapoff[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) tapoff[t] = (s == 2 ? (unsigned)((t & 1) * PL + (t >> 1)) : (unsigned)t) * 64u;

    if (total > 0) {
        gload(0);
        lstore();
        __syncthreads();
        for (int sidx = 0; sidx < total; ++sidx) {
            if (sidx + 1 < total) gload(sidx + 1);
            int xo0 = pt[0], xo1 = pt[4];
            for (int t = 0; t < nks; ++t) {
                const lds_u8* xb0 = Xb + xo0;
                const lds_u8* xb1 = Xb + xo1;
                if (t + 1 < nks) { xo0 = pt[16 * (t + 1)]; xo1 = pt[16 * (t + 1) + 4]; }
                const s16x4 a0 = tr_read(Ga + t * 1024), a1 = tr_read(Ga + t * 1024 + 256);
                const wb_bf16x8 a = __builtin_bit_cast(wb_bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) {
                    const s16x4 b0 = tr_read(xb0 + tapoff[tt]), b1 = tr_read(xb1 + tapoff[tt]);
                    const wb_bf16x8 bb = __builtin_bit_cast(wb_bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[tt], 0, 0, 0);
                }
            }
            __syncthreads();
            if (sidx + 1 < total) {
                lstore();
                __syncthreads();
            }
        }
    }

    // epilogue: partial slab [split][ky*k + kx][Cg][Cx]; wave (wr, wc) holds G channels wr*32.., X channels wc*32.., tap kx = tile
    const int KK = p.k * p.k;
    const int cx = cx0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (t >= p.k) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cg = cg0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            p.part[(((size_t)split * KK + ky * p.k + t) * p.Cg + cg) * p.Cx + cx] = acc[t][r];
        }
    }
}

// Waves per SIMD the compiler is asked to fit.  Left alone it hoists every fragment read of a k-step and, for the 7- and
// 9-tile variants, lands at 300+ registers = ONE wave per SIMD; within the budget (no spills) the extra resident
// workgroups hide the LDS/barrier latency these short k-steps cannot: 9x9 586 -> 948, 7x7 616 -> 834 TFLOP/s.
// The large staging class (XV 8) is limited to two workgroups per CU by its ~55 KB of LDS anyway.
template <int NTW, int XV>
constexpr int wb_min_waves() {
    if (XV >= 8) return NTW >= 7 ? 2 : 1;
    if (XV == 4 && NTW >= 7) return 2;          // (three would spill)
    return NTW >= 9 ? 2 : NTW >= 4 ? 3 : NTW == 3 ? 4 : 1;
}

template <int NTW, int XV, int GV>
__global__ __launch_bounds__(256, (wb_min_waves<NTW, XV>())) void conv_wgrad_bf16(const WgradBfParams p) {
    wgrad_bf16_body<NTW, XV, GV>(p);
}

// Sum the split-K slabs in a fixed order and scatter to the destination layout (same contract as
// wgrad_reduce_kernel of conv_wgrad.hip).
__global__ void wgrad_bf16_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int KK, int R,
                                         int C, int ld, int off, int transpose, int flip) {
    const int64_t n = (int64_t)KK * R * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float sum = 0.f;
        // eight slabs per trip, loaded together (from clamped addresses) and added in slab order: the launches with few outputs
        // and hundreds of slabs (16-61 workgroups) were one dependent load at a time, 120-150 us for a few KB of gradient
        for (int sp = 0; sp < S; sp += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(sp + q < S ? sp + q : S - 1) * n + i];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (sp + q < S) sum += v[q];
        }
        const int c = (int)(i % C);
        const int64_t t2 = i / C;
        const int r = (int)(t2 % R);
        int tap = (int)(t2 / R);
        if (flip) tap = KK - 1 - tap;
        if (transpose) dw[((size_t)tap * C + c) * ld + off + r] = sum;
        else dw[((size_t)tap * R + r) * ld + off + c] = sum;
    }
}

namespace {

struct PlanBf {
    WgradBfParams P;
    int transpose;
    int cls;            // staging class: 1 small (XV 3, GV 1), 2 large (XV 8, GV 7)
    size_t ws_bytes, lds_bytes;
    int blocks;
};

size_t lds_need(int nks, int nr, int rp) { return (size_t)nks * 16 * 128 + (size_t)nr * rp * 128 + (size_t)nks * 16 * 4 + 1024; }

// cfg: 0 automatic, 1 force the small staging class (32-pixel pieces of one row), 2 force the large one
bool make_plan_bf(const gdn_conv_geom* g, int Cx_in, int cfg, PlanBf& pl) {
    const bool auto_cfg = cfg == 0;
    if (!g || g->k < 1 || g->k > WB_KMAX || g->stride < 1 || g->stride > 2) return false;
    int Ho, Wo;
    if (gdn_conv_out_dims(g, &Ho, &Wo) != GDN_OK) return false;
    WgradBfParams& P = pl.P;
    P = WgradBfParams{};
    P.B = g->B; P.k = g->k; P.stride = g->stride; P.pad = g->pad; P.pad_mode = g->pad_mode;
    pl.transpose = 0;
    if (!g->transposed) {
        P.Hg = Ho; P.Wg = Wo; P.Cg = g->Cout;            // G role: dy
        P.Hx = g->H; P.Wx = g->W; P.Cx = Cx_in;          // X role: x
    } else {
        // ConvTranspose2d: dW[ci][co][tap] = sum_i X[i][ci] * dY[i*s - p + tap][co]
        P.Hg = g->H; P.Wg = g->W; P.Cg = Cx_in;          // G role: x (layer input)
        P.Hx = Ho; P.Wx = Wo; P.Cx = g->Cout;            // X role: dy
        pl.transpose = 1;
    }
    if ((P.Cg % 64) || (P.Cx % 64)) return false;
    const int s = P.stride, k = P.k;
    // ---- segment geometry ----
    // staging classes: 1 small (32-pixel pieces of one row: XV 3, GV 1), 3 medium (<= 112 flattened pixels: XV 4, GV 4,
    // ~28 KB of LDS -> more workgroups per CU), 2 large (<= 208 pixels: XV 8, GV 7, ~55 KB of LDS)
    int tw = 0, nr = 1;
    pl.cls = 2;
    // measured (profiles/r01_tune_conv_bf16_v4.txt): the medium class wins on the small levels (W <= 52: 3x3 @16x52
    // 575 -> 644, @8x26 213 -> 355 TFLOP/s) and on 3x3 stride-1 at W = 104; wider rows / larger k prefer 208-pixel segments
    if (cfg == 0 && (P.Wg <= 52 || (P.Wg <= 104 && s == 1 && k <= 3))) cfg = 3;
    const int segmax = cfg == 3 ? 112 : WB_SEGMAX, xvmax = cfg == 3 ? 4 : 8;
    if (cfg == 3) pl.cls = 3;
    if (P.Wg <= segmax && cfg != 1) {
        // whole rows: as many as fit the staging class and ~60 KB of LDS
        const int npos = (P.Wg - 1) * s + k, xl = cdiv(npos, 32), rp = s == 2 ? 2 * ((npos + 1) / 2) : npos;
        int n = segmax / P.Wg;
        if (n > 8) n = 8;
        if (n > P.Hg) n = P.Hg;
        while (n >= 1 && (n * xl > xvmax || lds_need(cdiv(n * P.Wg, 16), n, rp) > 60 * 1024)) --n;
        if (n >= 1) { tw = P.Wg; nr = n; }
    }
    if (tw == 0) {
        // a row that must be cut into pieces: for stride-2 layers the 104-pixel pieces of the large class (7 X passes per
        // piece, 7 k-steps) run 2.6x slower than 32-pixel pieces of the small class (measured: 128 vs 325-337 TFLOP/s)
        if (cfg == 1 || (auto_cfg && s == 2)) { tw = 32; pl.cls = 1; }
        else {
            // widest piece of one row the staging class holds, best row coverage first (multiples of 16, or an exact
            // divisor of the row such as 104 = 208/2)
            double best_eff = 0.0;
            for (int c = segmax; c >= 16; --c) {
                if ((c % 16) && (P.Wg % c)) continue;
                const int npos = (c - 1) * s + k, rp = s == 2 ? 2 * ((npos + 1) / 2) : npos;
                if (cdiv(npos, 32) > xvmax || lds_need(cdiv(c, 16), 1, rp) > 60 * 1024) continue;
                const double eff = (double)P.Wg / (double)(cdiv(P.Wg, c) * cdiv(c, 16) * 16);
                if (eff > best_eff + 1e-9) { best_eff = eff; tw = c; }
            }
            if (tw == 0) return false;
        }
    }
    P.tw = tw; P.nr = nr;
    P.npos = (tw - 1) * s + k;
    P.xl = cdiv(P.npos, 32);
    P.pl = (P.npos + 1) / 2;
    P.rp = s == 2 ? 2 * P.pl : P.npos;
    P.nks = cdiv(nr * tw, 16);
    if (P.nks * 16 > WB_SEGMAX) return false;
    const int xv = nr * P.xl, gv = cdiv(P.nks * 16, 32);
    if (pl.cls == 1 && (xv > 3 || gv > 1)) return false;
    if (pl.cls == 2 && (xv > 8 || gv > 7)) return false;
    if (pl.cls == 3 && (xv > 4 || gv > 4)) return false;
    pl.lds_bytes = lds_need(P.nks, nr, P.rp);
    if (pl.lds_bytes > 64 * 1024) return false;
    P.n_cgt = P.Cg / 64; P.n_cxt = P.Cx / 64;
    const int base = k * P.n_cgt * P.n_cxt;
    const int R = P.B * P.Hg;
    // split-K over image rows: whole segments per split, ~2-4 rounds of resident workgroups
    {
        const int occ = 2, slots = 256 * occ;
        const int nsegs_total = cdiv(R, nr);
        int bestS = 1; double best = -1.0;
        const int smax = nsegs_total < 512 ? nsegs_total : 512;
        for (int S = 1; S <= smax; ++S) {
            const int sps = cdiv(nsegs_total, S);               // segments (of nr rows) per split
            const int Se = cdiv(nsegs_total, sps);
            const double fill = (double)base * Se / slots;
            if (fill > 4.0 && best > 0) break;
            const double rounds = fill < 1.0 ? 1.0 : (double)cdiv(base * Se, slots);
            double eff = fill / rounds;
            eff *= (double)nsegs_total / ((double)sps * Se);
            eff *= 1.0 - 0.15 / rounds;
            if (eff > best + 0.01) { best = eff; bestS = S; }
        }
        const int sps = cdiv(nsegs_total, bestS);
        P.rows_per_split = sps * nr;
        P.S = cdiv(R, P.rows_per_split);
    }
    pl.blocks = base * cdiv(P.S, 8) * 8;
    pl.ws_bytes = (size_t)P.S * k * k * P.Cg * P.Cx * sizeof(float);
    return true;
}

template <int XV, int GV>
int launch_cls(const WgradBfParams& P, int blocks, size_t lds, hipStream_t st) {
    const dim3 grid(blocks), blk(256);
#define WB_LAUNCH(N) hipLaunchKernelGGL((conv_wgrad_bf16<N, XV, GV>), grid, blk, lds, st, P)
    switch (P.k) {
        case 1: WB_LAUNCH(1); break;
        case 2: case 3: WB_LAUNCH(3); break;
        case 4: WB_LAUNCH(4); break;
        case 5: WB_LAUNCH(5); break;
        case 6: case 7: WB_LAUNCH(7); break;
        default: WB_LAUNCH(9); break;
    }
#undef WB_LAUNCH
    return gdn_launch_status();
}

// ---- wgrad_ring_bf16 (wgrad_ring.h): plan ----
struct PlanRing {
    WgRingParams P;
    size_t ws_bytes;
    int blocks;
};

// Which layers take the ring kernel: stride-1 Conv2d with a 5x5 ... 9x9 window (3x3: two filter rows x three taps of a
// 64 x 64 tile are too little MFMA work per staged byte -- those layers keep the round-1 kernel until a wider tile exists).
bool make_plan_ring(const gdn_conv_geom* g, int Cx_in, bool forced, PlanRing& pl) {
    if (!g || g->transposed || g->stride != 1) return false;
    const int k = g->k;
    if (k != 3 && k != 5 && k != 7 && k != 9) return false;
    if (!forced && k == 3) return false;
    int Ho, Wo;
    if (gdn_conv_out_dims(g, &Ho, &Wo) != GDN_OK) return false;
    if ((g->Cout % 64) || (Cx_in % 64) || Wo < 9 || Ho < 1) return false;
    if (g->pad_mode == 1 && (g->pad >= g->H || g->pad >= g->W)) return false;
    WgRingParams& P = pl.P;
    P = WgRingParams{};
    P.B = g->B; P.k = k; P.pad = g->pad; P.pad_mode = g->pad_mode;
    P.Hg = Ho; P.Wg = Wo; P.Cg = g->Cout;
    P.Hx = g->H; P.Wx = g->W; P.Cx = Cx_in;
    // stage geometry: whole rows while they fit 224 pixels (as many as the X images allow), else 16-aligned strips of <= 208
    if (Wo <= WR_GPX) {
        P.nstrip = 1; P.TW = Wo; P.TWp = (Wo + 7) & ~7;
        int n = WR_GPX / P.TWp;
        if (n > Ho) n = Ho;
        while (n > 1 && (size_t)2 * (n + 1) * (P.TWp + 8) * 128 > WR_XBYTES) --n;
        P.nr = n;
    } else {
        P.nstrip = cdiv(Wo, 208);
        P.TW = cdiv(cdiv(Wo, P.nstrip), 16) * 16;
        P.TWp = P.TW; P.nr = 1;
    }
    P.ring = P.nr == 1;
    P.nks = cdiv(P.nr * P.TWp, 16);
    // X row pitch: the last k-step's windows end 8 positions past its 16 pixels.  A ring row is read in whole k-steps, so a width
    // that is 8 mod 16 needs the pitch of the next multiple of 16 (the pixels past TWp meet a zero dY, but they must be LDS this
    // launch wrote: stale NaN bit patterns times zero are NaN -- tests/diag/wgrad_ring_poison.py)
    P.RP = (P.ring ? P.nks * 16 : P.TWp) + 8;
    if (P.nks * 16 > WR_GPX) return false;
    if ((size_t)(P.ring ? 3 : 2 * (P.nr + 1)) * P.RP * 128 > WR_XBYTES) return false;
    if ((P.ring ? 1 : P.nr + 1) * (P.RP / 8) > (P.ring ? 32 : 48)) return false;     // DMA pieces per stage the kernel unrolls
    P.sph = cdiv(Ho, P.nr);
    P.nst = P.B * P.nstrip * P.sph;
    P.n_cgt = P.Cg / 64; P.n_cxt = P.Cx / 64;
    P.gmagic = 65536 / P.TWp + 1; P.xmagic = 65536 / (P.RP / 8) + 1;
    // split-K over stages: one workgroup per CU, every split runs all (filter-row pair, channel tile) units side by side
    const int wps = ((k + 1) / 2) * P.n_cgt * P.n_cxt;
    int S = gdn_plan_cus(g) / wps;
    if (S < 1) S = 1;
    if (S > P.nst) S = P.nst;
    P.ips = cdiv(P.nst, S);
    P.S = cdiv(P.nst, P.ips);
    pl.blocks = cdiv(P.S * wps, 8) * 8;
    pl.ws_bytes = (size_t)P.S * k * k * P.Cg * P.Cx * sizeof(float);
    return true;
}

int launch_ring(const WgRingParams& P, int blocks, hipStream_t st) {
    const dim3 grid(blocks), blk(512);
#define WR_LAUNCH(K) do { if (P.ring) hipLaunchKernelGGL((wgrad_ring_bf16<K, true>), grid, blk, 0, st, P); \
                          else hipLaunchKernelGGL((wgrad_ring_bf16<K, false>), grid, blk, 0, st, P); } while (0)
    switch (P.k) {
        case 3: WR_LAUNCH(3); break;
        case 5: WR_LAUNCH(5); break;
        case 7: WR_LAUNCH(7); break;
        default: WR_LAUNCH(9); break;
    }
#undef WR_LAUNCH
    return gdn_launch_status();
}

}  // namespace

extern "C" size_t gdn_conv_wgrad_bf16_workspace_bytes(const gdn_conv_geom* g, int32_t Cx, int32_t cfg) {
    PlanRing pr;
    if (((cfg & 7) == 0 || (cfg & 7) == 4) && make_plan_ring(g, Cx, (cfg & 7) == 4, pr)) return pr.ws_bytes;
    if ((cfg & 7) == 4) return 0;
    PlanBf pl;
    if (!make_plan_bf(g, Cx, cfg & 3, pl)) return 0;
    return pl.ws_bytes;
}

extern "C" int gdn_conv_wgrad_bf16(const gdn_conv_geom* g, const void* x, int32_t ldx, int32_t Cx, const void* dy,
                                   int32_t ldy, float* dw, int32_t ld_dw, int32_t ci_off, void* workspace,
                                   size_t workspace_bytes, int32_t cfg, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other HIP users of this thread
    if (!x || !dy || !dw) return GDN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    PlanRing pr;
    if (((cfg & 7) == 0 || (cfg & 7) == 4) && make_plan_ring(g, Cx, (cfg & 7) == 4, pr)) {
        if (!workspace || workspace_bytes < pr.ws_bytes) return GDN_ERR_WORKSPACE;
        WgRingParams& R = pr.P;
        R.g = dy; R.ldg = ldy; R.x = x; R.ldx = ldx;
        if ((R.ldg % 8) || (R.ldx % 8) || ((uintptr_t)R.g % 16) || ((uintptr_t)R.x % 16)) return GDN_ERR_UNSUPPORTED;
        const uint64_t gb = (((uint64_t)R.B * R.Hg * R.Wg - 1) * (uint64_t)R.ldg + R.Cg) * 2;
        const uint64_t xb = (((uint64_t)R.B * R.Hx * R.Wx - 1) * (uint64_t)R.ldx + R.Cx) * 2;
        if (gb >= 0xFF000000ull || xb >= 0xFF000000ull) return GDN_ERR_UNSUPPORTED;
        R.g_bytes = (unsigned)gb; R.x_bytes = (unsigned)xb;
        R.part = (float*)workspace;
        R.knobs = (cfg >> 12) & 15;
        int rc = launch_ring(R, pr.blocks, st);
        if (rc != GDN_OK) return rc;
        const int KK = R.k * R.k;
        const int64_t n = (int64_t)KK * R.Cg * R.Cx;
        const int blocks = (int)(cdiv64(n, 256) < 2048 ? cdiv64(n, 256) : 2048);
        hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, R.S, KK,
                           R.Cg, R.Cx, ld_dw, ci_off, 0, 0);
        return gdn_launch_status();
    }
    if ((cfg & 7) == 4) return GDN_ERR_UNSUPPORTED;
    PlanBf pl;
    if (!make_plan_bf(g, Cx, cfg & 3, pl)) return GDN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < pl.ws_bytes) return GDN_ERR_WORKSPACE;
    WgradBfParams& P = pl.P;
    if (pl.transpose) { P.g = x; P.ldg = ldx; P.x = dy; P.ldx = ldy; }
    else { P.g = dy; P.ldg = ldy; P.x = x; P.ldx = ldx; }
    if ((P.ldg % 8) || (P.ldx % 8) || ((uintptr_t)P.g % 16) || ((uintptr_t)P.x % 16)) return GDN_ERR_UNSUPPORTED;
    P.part = (float*)workspace;
    {
        const uint64_t gb = (((uint64_t)P.B * P.Hg * P.Wg - 1) * (uint64_t)P.ldg + P.Cg) * 2;
        const uint64_t xb = (((uint64_t)P.B * P.Hx * P.Wx - 1) * (uint64_t)P.ldx + P.Cx) * 2;
        if (gb >= 0xFF000000ull || xb >= 0xFF000000ull) return GDN_ERR_UNSUPPORTED;
        P.g_bytes = (unsigned)gb; P.x_bytes = (unsigned)xb;
    }
    int rc = pl.cls == 1 ? launch_cls<3, 1>(P, pl.blocks, pl.lds_bytes, st)
           : pl.cls == 3 ? launch_cls<4, 4>(P, pl.blocks, pl.lds_bytes, st)
                         : launch_cls<8, 7>(P, pl.blocks, pl.lds_bytes, st);
    if (rc != GDN_OK) return rc;
    const int KK = P.k * P.k;
    const int64_t n = (int64_t)KK * P.Cg * P.Cx;
    const int blocks = (int)(cdiv64(n, 256) < 2048 ? cdiv64(n, 256) : 2048);
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, P.S, KK,
                       P.Cg, P.Cx, ld_dw, ci_off, pl.transpose, 0);
    return gdn_launch_status();
}
