// fp32 per-bin GEMMs as bf16 x 3 split products on v_mfma_f32_32x32x16_bf16 (see gemm_x3.h for the arithmetic and the layout).
#include "gemm_x3.h"


namespace {

// ---- packing: fp32 [bins][rows][K] -> panels.  thread = (row, 8 consecutive k): two 16-byte loads, three 16-byte stores ----
__global__ __launch_bounds__(256) void x3_pack_rows_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int rows,
                                                           int K, int rows_pad) {
    const int KB = K / X3_BK, k8n = K / 8;
    const int bin = blockIdx.y;
    const size_t per_bin = x3_packed_bytes(rows, K);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)rows_pad * k8n; i += (int64_t)gridDim.x * 256) {
        // a wave = 16 consecutive rows x the 4 chunks of one k block: 1 KB contiguous per plane in the packed layout
        const int row = (int)((i >> 2) % rows_pad), k8 = (int)(i & 3) + 4 * (int)((i >> 2) / rows_pad);
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (row < rows) {
            const float* s = src + ((size_t)bin * rows + row) * K + k8 * 8;
            lo = *reinterpret_cast<const f32x4*>(s);
            hi = *reinterpret_cast<const f32x4*>(s + 4);
        }
        unsigned h[3][4];
        x3_split2(lo[0], lo[1], h[0][0], h[1][0], h[2][0]);
        x3_split2(lo[2], lo[3], h[0][1], h[1][1], h[2][1]);
        x3_split2(hi[0], hi[1], h[0][2], h[1][2], h[2][2]);
        x3_split2(hi[2], hi[3], h[0][3], h[1][3], h[2][3]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 v = {h[p][0], h[p][1], h[p][2], h[p][3]};
            *reinterpret_cast<uint4*>(dst + (size_t)bin * per_bin + x3_off(row, k8 * 8, p, KB)) = v;
        }
    }
}

// One 32-deep stage of a wave's 64 x 64 tile from the LDS images As / Bs ([plane][128 rows][64 B], chunk-swizzled): two
// sub-steps of 6 + 6 fragment reads and 24 MFMAs.
__device__ __forceinline__ void x3_stage_mfma(const unsigned char* As, const unsigned char* Bs, int a_base, int b_base, int h, int f_sw,
                                              f32x16 (&acc)[2][2], f32x16 (&cor)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int co = ((2 * s + h) ^ f_sw) * 16;
        x3_bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[i][p] = __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(&As[p * 8192 + a_base + i * 2048 + co]));
                bf[i][p] = __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(&Bs[p * 8192 + b_base + i * 2048 + co]));
            }
        // The bf16 MFMA aligns its 16 products AND the C input to the largest exponent among them and truncates what
        // falls below ~half an ulp of it, per addend, toward zero (measured: tests/diag/mfma_rounding.py); the sum itself is
        // rounded to nearest.  A leading product a1*b1 has a 16-bit significand, so against a running sum sqrt(K) larger
        // it loses nothing; a correction product sits 8 or 16 bits lower and WOULD be cut at the running sum's ulp.  Hence
        // two accumulators: `acc` chains the leading products, `cor` chains the five correction products (its own ulp is
        // 2^-8 of acc's), and the two meet once, in the epilogue.  No vector-ALU work in the loop; 32 roundings per
        // K = 512 instead of the 512 of the k-ordered fp32 MFMA chain this replaces.
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], cor[i][j], 0, 0, 0);
                cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], cor[i][j], 0, 0, 0);
                cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], cor[i][j], 0, 0, 0);
                cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], cor[i][j], 0, 0, 0);
                cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], cor[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    }
}

// ---- the GEMM ----
// Workgroup = 128 x 128 output tile, 4 waves (2 x 2) of 64 x 64 = 2 x 2 MFMA tiles; a stage is 32 k: per wave 2 sub-steps of
// (6 + 6 fragment reads, 24 MFMAs).  LDS: three bf16 planes of A and of B, [plane][128 rows][64 B] each (48 KB); two
// workgroups per CU (232 registers: two accumulator sets).  A is staged global (fp32) -> registers -> split -> LDS with the loads of stage i + 1 issued
// before the MFMAs of stage i; B's stage is a linear 24 KB copy of its panel.
// XCD-aware order (as wino_gemm_kernel): XCD j owns bins j, j + 8, ... and walks them with the N tiles of one M tile back to
// back, so an A tile is fetched from the fabric once and a bin's weight panels stay in that XCD's L2.
__global__ __launch_bounds__(256, 2) void gemm_x3_nt_kernel(const float* __restrict__ A, const unsigned char* __restrict__ Bp,
                                                            float* __restrict__ C, int M, int N, int K, int bins) {
    __shared__ __attribute__((aligned(16))) unsigned char As[3 * 128 * 64], Bs[3 * 128 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int NT = N / 128, MT = (M + 127) / 128, KB = K / X3_BK;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (NT * MT)) * 8 + xcd;
    if (bin >= bins) return;
    const int m0 = ((sq / NT) % MT) * 128, nt = sq % NT;
    const float* Ab = A + (size_t)bin * M * K;
    const unsigned char* Bb = Bp + (size_t)bin * x3_packed_bytes(N, K) + (size_t)nt * KB * (3 * 128 * 64);
    float* Cb = C + (size_t)bin * M * N;

    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    // A staging: 8 lanes fetch one row's 32 k = a full 128-byte line (lane = 4 k), a wave instruction 8 rows; a thread holds
    // rows rb, rb + 32, rb + 64, rb + 96.  (Fetching 16 k per thread -- 16-byte pieces of 32 different lines per instruction --
    // measured 1.2x slower: the texture path, not the matrix pipe, set the pace.)  Rows past M read row M - 1.
    const int k4 = tid & 7, rb0 = tid >> 3;
    const float* ap[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = v * 32 + rb0;
        ap[v] = Ab + (size_t)(m0 + row < M ? m0 + row : M - 1) * K + k4 * 4;
    }
    f32x4 ra[4];
    f32x4 rb[6];
    auto gload = [&](int kb) {
#pragma unroll
        for (int v = 0; v < 4; ++v) ra[v] = *reinterpret_cast<const f32x4*>(ap[v] + (size_t)kb * X3_BK);
        const unsigned char* bsrc = Bb + (size_t)kb * (3 * 128 * 64) + tid * 16;
#pragma unroll
        for (int v = 0; v < 6; ++v) rb[v] = *reinterpret_cast<const f32x4*>(bsrc + v * 4096);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = v * 32 + rb0;
            unsigned h0[3], h1[3];
            x3_split2(ra[v][0], ra[v][1], h0[0], h0[1], h0[2]);
            x3_split2(ra[v][2], ra[v][3], h1[0], h1[1], h1[2]);
            const int off = row * 64 + (((k4 >> 1) ^ x3_sw(row)) * 16) + (k4 & 1) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint2 w = {h0[p], h1[p]};
                *reinterpret_cast<uint2*>(&As[p * 8192 + off]) = w;
            }
        }
#pragma unroll
        for (int v = 0; v < 6; ++v) *reinterpret_cast<f32x4*>(&Bs[v * 4096 + tid * 16]) = rb[v];
    };
    // fragment addresses: lane (r = lane & 31, h = lane >> 5) reads 16 B of row (tile row base + r) at chunk (2 s + h) ^ swz(row)
    const int r32 = lane & 31, h = lane >> 5;
    const int f_sw = x3_sw(r32);                                       // tile row bases are multiples of 32: same swizzle
    const int a_base = (wm * 64 + r32) * 64, b_base = (wn * 64 + r32) * 64;

    gload(0);
    for (int kb = 0; kb < KB; ++kb) {
        lstore();
        __syncthreads();
        if (kb + 1 < KB) gload(kb + 1);
        x3_stage_mfma(As, Bs, a_base, b_base, h, f_sw, acc, cor);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = nt * 128 + wn * 64 + j * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) Cb[(size_t)m * N + col] = acc[i][j][r] + cor[i][j][r];
            }
        }
}

// ---- NT kernel, second structure: 512 threads, double-buffered LDS, ONE barrier per stage ----
// 8 waves (2 x 4): wave tile (TM/2) x 32 = MI x 1 MFMA tiles (TM = 128: MI = 2; TM = 64 for GEMMs with few rows: MI = 1).  Both
// operands' stage k + 1 is written to the other LDS buffer while stage k is multiplied, so the split arithmetic and the LDS
// writes of one wave run under the MFMAs of the wave it shares a SIMD with, and a stage costs one barrier:
//     iteration k:  split(regs) -> buf[(k+1)&1];  global loads of stage k + 2 -> regs;  MFMAs from buf[k&1];  barrier
// LDS 2 x (3 TM + 3 x 128) x 64 B = 96 KB at TM = 128: one workgroup (two waves per SIMD) per CU.
template <int TM>
__global__ __launch_bounds__(512, 2) void gemm_x3_nt8_kernel(const float* __restrict__ A, const unsigned char* __restrict__ Bp,
                                                             float* __restrict__ C, int M, int N, int K, int bins) {
    constexpr int MI = TM / 64, A_BYTES = 3 * TM * 64, B_BYTES = 3 * 128 * 64;
    __shared__ __attribute__((aligned(16))) unsigned char As[2][A_BYTES], Bs[2][B_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int NT = N / 128, MT = (M + TM - 1) / TM, KB = K / X3_BK;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (NT * MT)) * 8 + xcd;
    if (bin >= bins) return;
    const int m0 = ((sq / NT) % MT) * TM, nt = sq % NT;
    const float* Ab = A + (size_t)bin * M * K;
    const unsigned char* Bb = Bp + (size_t)bin * x3_packed_bytes(N, K) + (size_t)nt * KB * B_BYTES;
    float* Cb = C + (size_t)bin * M * N;
    f32x16 acc[MI], cor[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; cor[i][r] = 0.f; }
    // A staging: 8 lanes per row (a full 128-byte line), 64 rows per pass, MI passes
    const int k4 = tid & 7, rb0 = tid >> 3;
    const float* ap[MI];
#pragma unroll
    for (int v = 0; v < MI; ++v) {
        const int row = v * 64 + rb0;
        ap[v] = Ab + (size_t)(m0 + row < M ? m0 + row : M - 1) * K + k4 * 4;
    }
    f32x4 ra[MI], rb[3];
    auto gload = [&](int kb) {
#pragma unroll
        for (int v = 0; v < MI; ++v) ra[v] = *reinterpret_cast<const f32x4*>(ap[v] + (size_t)kb * X3_BK);
        const unsigned char* bsrc = Bb + (size_t)kb * B_BYTES + tid * 16;
#pragma unroll
        for (int v = 0; v < 3; ++v) rb[v] = *reinterpret_cast<const f32x4*>(bsrc + v * 8192);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int v = 0; v < MI; ++v) {
            const int row = v * 64 + rb0;
            unsigned h0[3], h1[3];
            x3_split2(ra[v][0], ra[v][1], h0[0], h0[1], h0[2]);
            x3_split2(ra[v][2], ra[v][3], h1[0], h1[1], h1[2]);
            const int off = row * 64 + (((k4 >> 1) ^ x3_sw(row)) * 16) + (k4 & 1) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint2 w = {h0[p], h1[p]};
                *reinterpret_cast<uint2*>(&As[buf][p * TM * 64 + off]) = w;
            }
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) *reinterpret_cast<f32x4*>(&Bs[buf][v * 8192 + tid * 16]) = rb[v];
    };
    const int r32 = lane & 31, h = lane >> 5;
    const int f_sw = x3_sw(r32);
    const int a_base = (wm * (TM / 2) + r32) * 64, b_base = (wn * 32 + r32) * 64;
    gload(0);
    lstore(0);
    if (KB > 1) gload(1);
    __syncthreads();
    for (int kb = 0; kb < KB; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < KB) lstore(cur ^ 1);                    // stage kb + 1 (in registers since the last iteration)
        if (kb + 2 < KB) gload(kb + 2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = ((2 * s + h) ^ f_sw) * 16;
            x3_bf16x8 af[MI][3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bf[p] = __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(&Bs[cur][p * 8192 + b_base + co]));
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i][p] = __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(&As[cur][p * TM * 64 + a_base + i * 2048 + co]));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {            // (accumulator roles: see x3_stage_mfma)
                cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], cor[i], 0, 0, 0);
                cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], cor[i], 0, 0, 0);
                cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], cor[i], 0, 0, 0);
                cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], cor[i], 0, 0, 0);
                cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], cor[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int col = nt * 128 + wn * 32 + r32;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) Cb[(size_t)m * N + col] = acc[i][r] + cor[i][r];
        }
}

// Reduction-over-rows GEMM (the weight gradients):  P[split][bin][i][j] = sum_t A[bin][t][i] * Bm[bin][t][j], t in the split's
// chunk of the T rows -- both operands fp32 row-major [bins][T][NI] / [bins][T][NJ], read as they lie (rows = the reduction).
// The MFMA wants 8 consecutive t per lane: a thread loads 8 rows x 2 channels (coalesced 8-byte loads, a wave covers 128
// channels of one row), so the 8 t values of a channel already sit in one thread's registers -- the transposition is the
// register naming -- splits them and writes one 16-byte chunk per plane and channel: the LDS image is the NT kernel's
// ([plane][channel][32 t], chunk-swizzled) and the MFMA stage is shared.  Both operands are split on the fly here (32 values
// per thread and stage, ~176 vector instructions under 48 MFMAs per wave).
__global__ __launch_bounds__(256, 2) void gemm_x3_tn_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            float* __restrict__ P, int T, int NI, int NJ, int nsplit, int bins) {
    __shared__ __attribute__((aligned(16))) unsigned char As[3 * 128 * 64], Bs[3 * 128 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int TI = NI / 128, TJ = NJ / 128;
    const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
    const int bin = (sq / (TI * TJ * nsplit)) * 8 + xcd;
    if (bin >= bins) return;
    const int split = (sq / (TI * TJ)) % nsplit, i0 = ((sq / TJ) % TI) * 128, j0 = (sq % TJ) * 128;
    const int chunk = ((T + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int tb = split * chunk, te = tb + chunk < T ? tb + chunk : T;        // this workgroup reduces rows [tb, te)
    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }
    // staging: thread = (t8 = tid >> 6: rows t8*8 .. t8*8+7 of the stage, cp = tid & 63: channels cp and cp + 64) -- a wave
    // instruction reads 64 consecutive channels of one row (256 B), and the eight lanes of a ds_write_b128 group write eight
    // CONSECUTIVE LDS rows at one chunk: conflict-free under x3_sw (with channels 2 cp, 2 cp + 1 per thread -- 8-byte loads --
    // the groups wrote even rows only: two-way conflicts on a third of all LDS cycles, profiles/r03_gemm_x3_pmc.json)
    const int t8 = tid >> 6, cp = tid & 63;
    const float* ap = A + ((size_t)bin * T) * NI + i0 + cp;
    const float* bp = Bm + ((size_t)bin * T) * NJ + j0 + cp;
    float ra[2][8], rb[2][8];
    auto gload = [&](int t0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = t0 + t8 * 8 + e;
            const bool in = t < te;
            const int tc = t < T ? t : T - 1;               // rows past the chunk are read (and dropped) from a row that exists
            const size_t ta = (size_t)tc * NI, tbo = (size_t)tc * NJ;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float va = ap[ta + c * 64], vb = bp[tbo + c * 64];
                ra[c][e] = in ? va : 0.f;
                rb[c][e] = in ? vb : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int c = 0; c < 2; ++c) {                    // the thread's two channels = two LDS rows
            const int row = cp + c * 64;
            const int off = row * 64 + ((t8 ^ x3_sw(row)) * 16);
            unsigned ha[3][4], hb[3][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x3_split2(ra[c][2 * q], ra[c][2 * q + 1], ha[0][q], ha[1][q], ha[2][q]);
                x3_split2(rb[c][2 * q], rb[c][2 * q + 1], hb[0][q], hb[1][q], hb[2][q]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 va = {ha[p][0], ha[p][1], ha[p][2], ha[p][3]}, vb = {hb[p][0], hb[p][1], hb[p][2], hb[p][3]};
                *reinterpret_cast<uint4*>(&As[p * 8192 + off]) = va;
                *reinterpret_cast<uint4*>(&Bs[p * 8192 + off]) = vb;
            }
        }
    };
    const int r32 = lane & 31, h = lane >> 5;
    const int f_sw = x3_sw(r32);
    const int a_base = (wm * 64 + r32) * 64, b_base = (wn * 64 + r32) * 64;
    gload(tb);
    for (int t0 = tb; t0 < te; t0 += 32) {
        lstore();
        __syncthreads();
        if (t0 + 32 < te) gload(t0 + 32);
        x3_stage_mfma(As, Bs, a_base, b_base, h, f_sw, acc, cor);
        __syncthreads();
    }
    float* Pb = P + (((size_t)split * bins + bin) * NI) * NJ;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j0 + wn * 64 + j * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Pb[(size_t)row * NJ + col] = acc[i][j][r] + cor[i][j][r];
            }
        }
}

}  // namespace

void launch_x3_pack_rows(const float* src, void* dst, int bins, int rows, int K, hipStream_t st) {
    const int rows_pad = (rows + X3_TILE - 1) / X3_TILE * X3_TILE;
    const int64_t n = (int64_t)rows_pad * (K / 8);
    const int64_t nb = cdiv64(n, 256);
    hipLaunchKernelGGL(x3_pack_rows_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096), bins), dim3(256), 0, st, src,
                       (unsigned char*)dst, rows, K, rows_pad);
}

void launch_gemm_x3_nt(const float* A, const void* Bp, float* C, int bins, int M, int N, int K, hipStream_t st) {
    const int NT = N / 128, MT = (M + 127) / 128, bg = (bins + 7) / 8;
    // Measured (B = 20, profiles/r03_gemm_x3_time_*.txt): the 256-thread kernel (two workgroups per CU, two barriers per stage)
    // is the fastest on the large GEMMs (level 3: 0.264 ms against 0.281 for the 512-thread double-buffered one); with few row
    // tiles (level 4, M = 1040: 576 tiles of 128 rows = 2.25 rounds of the chip) 64-row tiles win (0.067 against 0.078 ms).
    // (the 512-thread 128-row variant gemm_x3_nt8_kernel<128> was measured against these two and is not dispatched)
    const bool small = (int64_t)MT * NT * bins < 4 * 256;
    if (small)
        hipLaunchKernelGGL(gemm_x3_nt8_kernel<64>, dim3(((M + 63) / 64) * NT * bg * 8), dim3(512), 0, st, A, (const unsigned char*)Bp, C, M,
                           N, K, bins);
    else
        hipLaunchKernelGGL(gemm_x3_nt_kernel, dim3(MT * NT * bg * 8), dim3(256), 0, st, A, (const unsigned char*)Bp, C, M, N, K, bins);
}

void launch_gemm_x3_tn(const float* A, const float* Bm, float* P, int bins, int T, int NI, int NJ, int nsplit, hipStream_t st) {
    const int bg = (bins + 7) / 8;
    hipLaunchKernelGGL(gemm_x3_tn_kernel, dim3((NI / 128) * (NJ / 128) * nsplit * bg * 8), dim3(256), 0, st, A, Bm, P, T, NI, NJ,
                       nsplit, bins);
}

// ---- C ABI: measurement / test hooks (the product path calls the launchers from the Winograd entry points) ----
extern "C" size_t gdn_gemm_x3_packed_bytes(int32_t bins, int32_t rows, int32_t K) {
    if (bins < 1 || rows < 1 || K < X3_BK || (K % X3_BK)) return 0;
    return (size_t)bins * x3_packed_bytes(rows, K);
}

extern "C" int gdn_gemm_x3_pack(const float* src, void* dst, int32_t bins, int32_t rows, int32_t K, void* stream) {
    (void)hipGetLastError();
    if (!src || !dst || bins < 1 || rows < 1 || K < X3_BK || (K % X3_BK)) return GDN_ERR_BAD_ARG;
    launch_x3_pack_rows(src, dst, bins, rows, K, (hipStream_t)stream);
    return gdn_launch_status();
}

extern "C" int gdn_gemm_x3_nt(const float* A, const void* Bp, float* C, int32_t bins, int32_t M, int32_t N, int32_t K,
                              void* stream) {
    (void)hipGetLastError();
    if (!A || !Bp || !C || bins < 1) return GDN_ERR_BAD_ARG;
    if (!gemm_x3_ok(M, N, K)) return GDN_ERR_UNSUPPORTED;
    launch_gemm_x3_nt(A, Bp, C, bins, M, N, K, (hipStream_t)stream);
    return gdn_launch_status();
}

// host query: the split count the reduction GEMM's callers use (gemm_x3.h cost model)
extern "C" int64_t gdn_gemm_x3_tn_splits(int32_t bins, int32_t T, int32_t NI, int32_t NJ) {
    if (bins < 1 || !gemm_x3_tn_ok(T, NI, NJ)) return 0;
    return gemm_x3_tn_splits(bins, T, NI, NJ);
}

extern "C" int gdn_gemm_x3_tn(const float* A, const float* Bm, float* P, int32_t bins, int32_t T, int32_t NI, int32_t NJ,
                              int32_t nsplit, void* stream) {
    (void)hipGetLastError();
    if (!A || !Bm || !P || bins < 1 || T < 1 || nsplit < 1) return GDN_ERR_BAD_ARG;
    if (!gemm_x3_tn_ok(T, NI, NJ)) return GDN_ERR_UNSUPPORTED;
    launch_gemm_x3_tn(A, Bm, P, bins, T, NI, NJ, nsplit, (hipStream_t)stream);
    return gdn_launch_status();
}
