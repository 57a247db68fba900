// fp32 GEMMs on the bf16 matrix pipe: "bf16 x 3" split products (gfx950).
//
// The per-bin GEMMs of the Winograd (and frequency-domain) layers are fp32 and ran on v_mfma_f32_32x32x2_f32 at 0.6-0.7 of its
// 157 TFLOP/s peak -- there is no faster fp32 matrix instruction on gfx950 (no TF32 / xf32).  The bf16 pipe is 16x faster, and an
// fp32 number is EXACTLY the sum of three bf16 numbers (24 = 8 + 8 + 8 significand bits):
//     a = a1 + a2 + a3,  b = b1 + b2 + b3,
//     a*b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a b|)
// so six bf16 products (each exact in the fp32 accumulator) reproduce the fp32 product to its own rounding level at 6/16 of the
// matrix-pipe cycles.  Operand traffic: the streamed operand A stays fp32 in memory and is split while it is staged into LDS
// (4 VALU instructions per element, hidden under the other waves' MFMAs); the re-used operand B (weights) is split once per step
// into packed panels.
//
// Packed operand ("panel") layout, one per bin: rows in tiles of 128, k in blocks of 32:
//     [row tile][k block][plane 0..2][row 0..127][64 bytes = 32 bf16]
// with the four 16-byte chunks of a row XOR-permuted by x3_sw(row): the ds_read_b128 of an MFMA fragment (32 rows x 16
// bytes per lane half) then touches every bank once, and so does a ds_write_b128 of eight consecutive rows at one chunk (the
// transposing loader of the reduction GEMM).  A (row tile, k block) stage is 24 KB contiguous: staged by a linear copy.
#pragma once
#include "common.h"

#define X3_TILE 128      // rows per panel / workgroup tile side
#define X3_BK 32         // reduction depth per stage

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));

// chunk permutation of a row (bits b4 .. b0 of the row): ((b1 ^ b4) << 1) | b2.  Reads: the 16 lanes of a ds_read_b128 group hold
// the row quads {0,3,5,6} or {1,2,4,7} of 32 rows -- four distinct values for every (b1, b0).  Writes of rows r .. r + 7 at one
// chunk: bijective in (b2, b1), so with b0 the eight 16-byte slots of a 128-byte bank window are all different.
__host__ __device__ static inline int x3_sw(int row) { return ((((row >> 1) ^ (row >> 4)) & 1) << 1) | ((row >> 2) & 1); }
// byte offset of element (row, k) of plane p inside one bin's packed operand (KB = K / 32 k blocks)
__host__ __device__ static inline size_t x3_off(int row, int k, int p, int KB) {
    const int t = row >> 7, r = row & 127, kb = k >> 5, ch = ((k >> 3) & 3) ^ x3_sw(r);
    return ((((size_t)t * KB + kb) * 3 + p) * 128 + r) * 64 + ch * 16 + (k & 7) * 2;
}
// bytes of one bin's packed operand with `rows` rows (padded to whole tiles) and reduction depth K
__host__ __device__ static inline size_t x3_packed_bytes(int rows, int K) {
    return (size_t)((rows + X3_TILE - 1) / X3_TILE) * (K / X3_BK) * 3 * 128 * 64;
}

// Two fp32 values -> three dwords of packed bf16 pairs (x in bits 0..15): x = x1 + x2 + x3 and y = y1 + y2 + y3 EXACTLY.
// Every term is a round-to-nearest-even bf16 of what is left (v_cvt_pk_bf16_f32); each remainder is exactly representable
// (24 -> 16 -> 8 significant bits).  Rounding, not truncating, matters: with truncated terms every correction product has the
// sign of the full product, and the matrix pipe's own truncation of the (small) correction sum against the (large) leading
// products then shrinks every result toward zero -- a coherent bias that survives sums over a million outputs (measured on the
// exactly-zero gradient of a BatchNorm bias, tests/diag/grad_accuracy.py); rounded terms have signs of their own.
__device__ __forceinline__ unsigned x3_cvt2(float lo, float hi) {
    typedef float x3_f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 x3_b2 __attribute__((ext_vector_type(2)));
    const x3_f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_b2));
}
__device__ __forceinline__ void x3_split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = x3_cvt2(x, y);
    const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
    p2 = x3_cvt2(rx, ry);
    const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
    p3 = x3_cvt2(sx, sy);
}

// C[bin][m][n] = sum_k A[bin][m][k] * B[bin][n][k]:  A fp32 row-major [bins][M][K]; Bp packed panels of B (x3_pack_rows);
// C fp32 row-major [bins][M][N].  N a multiple of 128, K a multiple of 32.
void launch_gemm_x3_nt(const float* A, const void* Bp, float* C, int bins, int M, int N, int K, hipStream_t st);
// packs fp32 row-major [bins][rows][K] into panels (rows padded with zeros to whole tiles)
void launch_x3_pack_rows(const float* src, void* dst, int bins, int rows, int K, hipStream_t st);
static inline bool gemm_x3_ok(int M, int N, int K) { return M >= 1 && N >= 128 && N % 128 == 0 && K >= 32 && K % 32 == 0; }
// P[split][bin][i][j] = sum over the split's rows t of A[bin][t][i] * Bm[bin][t][j]  (fp32 row-major operands [bins][T][NI] /
// [bins][T][NJ], both split on the fly); the caller sums the nsplit partial sets in a fixed order.  NI, NJ multiples of 128.
void launch_gemm_x3_tn(const float* A, const float* Bm, float* P, int bins, int T, int NI, int NJ, int nsplit, hipStream_t st);
static inline bool gemm_x3_tn_ok(int T, int NI, int NJ) { return T >= 1 && NI >= 128 && NI % 128 == 0 && NJ >= 128 && NJ % 128 == 0; }
// splits of the reduction.  512 workgroup slots (two per CU); chunks of at least 256 rows, at most 16 splits.  Cost model, in
// units of one whole-K tile: rounds of the chip / s for the GEMM, plus the extra partial-product sets the tap kernel has to read
// back (0.23 per set at T = 1040 and 36 x [512 x 512] products, measured: ~20 us against ~87 us per round; scaled by T and the set's size) -- the F(4x4,3x3) level-3 weight
// gradient has 576 tiles: one split = 2 rounds (175 us), two = 2.25 half rounds paid as 3 (1.5 + 0.23).  With fewer tiles than
// slots this reduces to the round-2 rule ceil(512 / tiles) (the 64 -> 128 stride-2 layer: 32 tiles over 30800 rows -> 16 splits).
static inline int gemm_x3_tn_splits(int bins, int T, int NI, int NJ) {
    const int wgs = (NI / 128) * (NJ / 128) * bins;
    int best = 1;
    double best_c = 1e30;
    for (int s = 1; s <= 16; ++s) {
        if (s > 1 && T / s < 256) break;
        const double c = (double)((wgs * s + 511) / 512) / s +
                         0.23 * (s - 1) * (1040.0 / T) * ((double)bins * NI * NJ / (36.0 * 512 * 512));
        if (c < best_c - 1e-9) { best_c = c; best = s; }
    }
    return best;
}
