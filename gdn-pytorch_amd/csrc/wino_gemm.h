// Shared by conv_wino.hip (F(2x2,3x3), stride-1 3x3 layers) and conv_wino2.hip (F(3x3,2x2), 4x4 stride-2 layers): the
// per-bin fp32 MFMA GEMMs of the Winograd domain and the pad-1 reflection fold.  Every translation unit that includes
// this header gets its own (anonymous-namespace) copy of the kernels.
#pragma once
#include "common.h"

#define WINO_BINS 16
#ifndef WINO_KC
#define WINO_KC 64
#endif
#ifndef GDN_KEEP
#define GDN_KEEP(v) asm volatile("" : "+v"(v))
#endif

namespace {

// thread = (tile, channel): block = 4 tiles x 64 channels, grid.x = (tile / 4) << q_shift | channel chunk
__device__ __forceinline__ bool wino_decode(int M, int q_shift, int& t, int& c) {
    t = (blockIdx.x >> q_shift) * 4 + (threadIdx.x >> 6);
    c = (blockIdx.x & ((1 << q_shift) - 1)) * 64 + (threadIdx.x & 63);
    return t < M;
}

// Per-bin real GEMM  Cm[bin][m][n] = sum_k A[bin][m][k] * Bm[bin][n][k]  (both K-contiguous), fp32 MFMA.
// 64x64 tile, 4 waves of one 32x32 MFMA tile, 64-wide k-steps (8 per GEMM at K = 512), the padded-pitch LDS image / b128 fragment scheme of
// conv_igemm_f32.  XCD-aware order: XCD j owns bins j and j + 8 and walks them bin-major with the N-tiles of one M-tile
// back to back (a bin's 1 MB weight matrix and every A tile stay in that XCD's L2).
template <int KC, int BM>      // KC reduction channels per step (32 / 64); BM rows per workgroup (64: one 32x32 MFMA tile
                               // per wave; 128: two, sharing the B fragment)
__global__ __launch_bounds__(256, BM == 64 ? 4 : 3) void wino_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                                          float* __restrict__ Cm, int M, int N, int K) {
    constexpr int LD = KC + 4, NV = KC / 32, RM = BM / 64;  // NV float4 per thread, row pass and operand; RM row tiles per wave
    __shared__ __attribute__((aligned(16))) float As[BM * LD], Bs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int NT = N / 64, MT = (M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (NT * MT)) * 8 + xcd, m0 = ((sq / NT) % MT) * BM, n0 = (sq % NT) * 64;
    const float* Ab = A + (size_t)bin * M * K;
    const float* Bb = Bm + (size_t)bin * N * K;
    float* Cb = Cm + (size_t)bin * M * N;
    // Two-level summation (as conv_igemm_f32): an MFMA accumulation is a strictly k-ordered fp32 fma chain, whose rounding
    // grows with sqrt(K) -- K reaches 2048 on the stride-2 layers.  Chains of 128 terms (two k-steps) start from C = 0 and are
    // folded into the running total.
    f32x16 acc[RM], run[RM], zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; run[t][r] = 0.f; }
    const int lr = tid >> 3, lc = (tid & 7) * 4;          // 32 rows per pass, 8 lanes x 16 B = one 32-float chunk per row
    f32x4 ra[2 * RM][NV], rb[2][NV];
    auto gload = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < 2 * RM; ++ps) {
            const int m = m0 + ps * 32 + lr;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                ra[ps][v] = m < M ? *reinterpret_cast<const f32x4*>(Ab + (size_t)m * K + k0 + v * 32 + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                rb[ps][v] = *reinterpret_cast<const f32x4*>(Bb + (size_t)(n0 + ps * 32 + lr) * K + k0 + v * 32 + lc);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2 * RM; ++ps)
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(&As[(ps * 32 + lr) * LD + v * 32 + lc]) = ra[ps][v];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(&Bs[(ps * 32 + lr) * LD + v * 32 + lc]) = rb[ps][v];
    };
    const int a_off = (wm * 32 * RM + (lane & 31)) * LD + (lane >> 5) * 16;     // wave rows: wm * 32 * RM + t * 32 + r
    const int b_off = (wn * 32 + (lane & 31)) * LD + (lane >> 5) * 16;
    gload(0);
    lstore();
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0 + KC < K) gload(k0 + KC);
        const bool fresh = (k0 % 128) == 0;                   // first k-step of a 128-term chain
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[b_off + v * 32 + g4 * 4]);
#pragma unroll
                for (int t = 0; t < RM; ++t) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&As[a_off + t * 32 * LD + v * 32 + g4 * 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (v == 0 && g4 == 0 && e == 0) {
                            if (fresh) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) run[t][r] += acc[t][r];
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], zero16, 0, 0, 0);
                            } else {
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[t], 0, 0, 0);
                            }
                        } else {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        __syncthreads();
        if (k0 + KC < K) { lstore(); __syncthreads(); }
    }
    const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < RM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 * RM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) Cb[(size_t)m * N + col] = run[t][r] + acc[t][r];
        }
}

// Reduction-over-tiles GEMM of the weight gradient:  P[bin][n][c] = sum_t Dv[bin][t][n] * V[bin][t][c]  (rows = tiles, both
// operands read as they lie).  64x64 output tile, 32 tiles of the reduction per step, conflict-free ds_read_b32 row reads.
// nsplit > 1: the reduction over the M tiles is cut into nsplit chunks, one workgroup each, writing partial products
// P[split][bin][NI][NJ] that the tap kernels sum in fixed order (layers with few channels have too few (bin, tile)
// workgroups to fill the chip otherwise: 128 for a 64 -> 128 stride-2 layer).
__global__ __launch_bounds__(256, 4) void wino_gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            float* __restrict__ P, int M, int NI, int NJ, int nsplit) {
    constexpr int LD = 64;
    __shared__ __attribute__((aligned(16))) float As[32 * LD], Bs[32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave >> 1, wj = wave & 1;
    const int TI = NI / 64, TJ = NJ / 64;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (TI * TJ * nsplit)) * 8 + xcd, split = (sq / (TI * TJ)) % nsplit;
    const int i0 = ((sq / TJ) % TI) * 64, j0 = (sq % TJ) * 64;
    const int mchunk = ((M + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int mb = split * mchunk, me = mb + mchunk < M ? mb + mchunk : M;         // this workgroup reduces tiles [mb, me)
    const float* Ab = A + (size_t)bin * M * NI + i0;
    const float* Bb = Bm + (size_t)bin * M * NJ + j0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    f32x4 ra[2], rb[2];
    auto gload = [&](int m0) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int m = m0 + ps * 16 + lr;
            ra[ps] = m < me ? *reinterpret_cast<const f32x4*>(Ab + (size_t)m * NI + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[ps] = m < me ? *reinterpret_cast<const f32x4*>(Bb + (size_t)m * NJ + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            *reinterpret_cast<f32x4*>(&As[(ps * 16 + lr) * LD + lc]) = ra[ps];
            *reinterpret_cast<f32x4*>(&Bs[(ps * 16 + lr) * LD + lc]) = rb[ps];
        }
    };
    const int a_off = (lane >> 5) * LD + wi * 32 + (lane & 31);
    const int b_off = (lane >> 5) * LD + wj * 32 + (lane & 31);
    gload(mb);
    lstore();
    __syncthreads();
    for (int m0 = mb; m0 < me; m0 += 32) {
        if (m0 + 32 < me) gload(m0 + 32);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[a_off + kk * LD], Bs[b_off + kk * LD], acc, 0, 0, 0);
        __syncthreads();
        if (m0 + 32 < me) { lstore(); __syncthreads(); }
    }
    float* Pb = P + ((size_t)split * WINO_BINS + bin) * NI * NJ;
    const int col = j0 + wj * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Pb[(size_t)i * NJ + col] = acc[r];
    }
}

// splits of the weight-gradient reduction: at least ~1024 workgroups, chunks of at least 256 tiles, at most 8
inline int wino_tn_splits(int M, int NI, int NJ) {
    const int wgs = (NI / 64) * (NJ / 64) * WINO_BINS;
    int s = (1024 + wgs - 1) / wgs;
    if (s > 8) s = 8;
    while (s > 1 && M / s < 256) --s;
    return s < 1 ? 1 : s;
}
inline void launch_wino_gemm_tn(const float* A, const float* Bm, float* P, int M, int NI, int NJ, int nsplit, hipStream_t st) {
    hipLaunchKernelGGL(wino_gemm_tn_kernel, dim3((NI / 64) * (NJ / 64) * WINO_BINS * nsplit), dim3(256), 0, st, A, Bm, P, M, NI, NJ, nsplit);
}

// 64-row tiles: the 128-row instantiation (two MFMA tiles per wave sharing the B fragment, 3 workgroups per CU) measured
// the same at level 3 (0.392 vs 0.397 ms per forward) and 8 % slower at level 4
// (and 7-23 % slower on the short-K GEMMs of the stride-2 layers, K = 128 / 256: 3.35 vs 3.14 ms over their eight forwards)
void launch_wino_gemm(const float* A, const float* Bm, float* Cm, int M, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL((wino_gemm_kernel<WINO_KC, 64>), dim3(cdiv(M, 64) * (N / 64) * WINO_BINS), dim3(256), 0, st, A, Bm, Cm, M, N, K);
}

// dx[y][x] = sum of the padded-domain gradient over the padded coordinates that reflect onto (y, x)  (+ addsrc), pad 1
__global__ __launch_bounds__(256) void wino_reflect_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int ldx,
                                                                const float* __restrict__ addsrc, int ld_add,
                                                                int B, int H, int W, int C) {
    const int Hp = H + 2, Wp = W + 2, c4n = C / 4;
    const int64_t total = (int64_t)B * H * W * c4n;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H), b = (int)(t / H);
        int qy[3], qx[3], ny = 0, nx = 0;
        qy[ny++] = y + 1;
        if (y == 1) qy[ny++] = 0;                  // padded row 0 mirrors row 1
        if (y == H - 2) qy[ny++] = H + 1;          // padded row H + 1 mirrors row H - 2
        qx[nx++] = x + 1;
        if (x == 1) qx[nx++] = 0;
        if (x == W - 2) qx[nx++] = W + 1;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e)
                s4 += *reinterpret_cast<const f32x4*>(dxp + ((size_t)(b * Hp + qy[a]) * Wp + qx[e]) * C + c4 * 4);
        const size_t op = (size_t)(b * H + y) * W + x;
        if (addsrc) s4 += *reinterpret_cast<const f32x4*>(addsrc + op * ld_add + c4 * 4);
        *reinterpret_cast<f32x4*>(dx + op * ldx + c4 * 4) = s4;
    }
}


}  // namespace
