// NOT PART OF THE LIBRARY (moved here in round 5): the LDS-DMA ring form of the bf16 x 3 GEMM, built and measured in round 4
// (profiles/r04d_x3_ring_time_and_clock.txt: 6-14 % faster back to back, 0-5 % between memory-bound kernels) and never wired
// into a layer.  Kept as a record of the experiment; it was reachable through gdn_gemm_x3_nt_packed (header revisions 214-218).
// fp32 per-bin GEMMs as bf16 x 3 split products, third structure: BOTH operands packed (gemm_x3.h panels), staged by LDS-DMA
// into a ring of k16 sub-stages, one workgroup per CU, persistent.  (Round 4; the structure of conv_ring.h applied to the NT GEMM.)
//
//   C[bin][m][n] = sum_k A[bin][m][k] * B[bin][n][k]     A, B: packed panels [row tile 128][k block 32][plane 3][row][64 B]
//
// * Work unit = 256 x 128 outputs (two A panels x one B panel; "type 0") or -- the odd last A panel of a bin -- 128 x 256 (one A
//   panel x two B panels; "type 1"): either way THREE panel slices per k16 sub-stage, 3 x (3 planes x 128 rows x 32 B) = 36 KB,
//   and 8 waves of 64 x 64 outputs (acc + cor: 128 accumulator registers, see x3_stage_mfma for the two-accumulator arithmetic).
// * Ring: 4 slots x 36 KB.  Sub-stage t + 3 is requested (LDS-DMA, buffer_load_dwordx4 ... lds) right after the barrier that opens
//   sub-stage t; that barrier also guarantees sub-stage t + 1 has landed (counted s_waitcnt vmcnt before it), so the fragments of
//   t + 1 are read AHEAD, under the MFMAs of t, and no wave starts a sub-stage by waiting for LDS.  ONE raw s_barrier per
//   sub-stage (24 MFMAs per wave), no vector-ALU work and no LDS writes in the loop.
// * LDS image of a slice: [plane][row][32 B]; the two 16-byte halves of a row are exchanged where (row >> 3) & 1: the 16 lanes
//   of a ds_read_b128 group ({0-3,12-15,20-27} ...) then touch every bank once.  The permutation is applied on the DMA's SOURCE
//   address (the panels keep the k32 layout of gemm_x3.h: sub-stage s of a k block is chunks (2s + half) ^ x3_sw(row)).
// * Persistent: 8 x wgx workgroups; XCD x owns bins x, x + 8, ... and walks their units with the N tiles of one A tile on
//   neighbouring workgroups (an A tile is fetched from the fabric once, a bin's B panels stay in that XCD's L2).  The units left
//   after the last full round are cut along K into up to 8 ranges on otherwise idle workgroups (fp32 slabs, summed in slab order by
//   x3r_combine_kernel): 2112 tiles on 512 slots were 4.125 rounds paid as 5.
// * The next unit's first three sub-stages are requested BEFORE the epilogue; the epilogue transposes 16 x 32 pieces through a
//   2 KB per-wave LDS scratch and leaves with 16-byte buffer stores that are issued unconditionally (rows past M are dropped by
//   the descriptor's bounds check), so the counted waits of the next unit know exactly how many younger stores are in flight.
#pragma once
#include "gemm_x3.h"

#define X3R_SLICE 12288                      // one panel's k16 slice: 3 planes x 128 rows x 32 B
#define X3R_SLOT (3 * X3R_SLICE)             // 36 KB
#define X3R_NSLOT 4
#define X3R_SCR (X3R_NSLOT * X3R_SLOT)       // epilogue scratch: 2 KB per wave
#define X3R_LDS (X3R_SCR + 8 * 2048)         // 163840 B = all 160 KB of a CU
#define X3R_UNIT_FLOATS 32768                // outputs of a unit (256 x 128 or 128 x 256)
#define X3R_PANEL_STAGE 24576                // bytes of one (row tile, k block) of a panel set
#define X3R_STORES 16                        // 16-byte stores per lane of a full epilogue

struct X3RingArgs {
    const unsigned char* Ap;                 // packed A [bins][PM panels][KB][3][128][64 B]
    const unsigned char* Bp;                 // packed B [bins][NTt panels][KB][3][128][64 B]
    float* C;                                // [bins][M][N]
    float* slabs;                            // [8 * wgx][X3R_UNIT_FLOATS] partial sums of the K-split tail units
    unsigned a_bytes, b_bytes;               // sizes of Ap / Bp (buffer descriptors address 32 bits)
    int bins, M, N, K;
    int PM, NTt;                             // panels per bin: cdiv(M, 128), N / 128
    int pairs, single, upb;                  // type-0 rows of units (PM / 2), PM & 1, units per bin
    int wgx;                                 // workgroups per XCD
    int max_parts;                           // K ranges a tail unit may be cut into (1: never split)
};

struct X3RingXcd { int units, rounds, tail, parts, sp; };        // sp: sub-stages per K range
struct X3RingUnit { int bin, type, row0, col0, pa0, pa1, pb0, pb1; };

__host__ __device__ static inline X3RingXcd x3r_xcd(const X3RingArgs& a, int xcd) {
    X3RingXcd r;
    const int nb = xcd < a.bins ? (a.bins - xcd + 7) / 8 : 0, KS = a.K / 16;
    r.units = nb * a.upb; r.rounds = r.units / a.wgx; r.tail = r.units - r.rounds * a.wgx;
    r.parts = 1; r.sp = KS;
    if (r.tail > 0 && 2 * r.tail <= a.wgx && a.max_parts > 1) {
        int parts = a.wgx / r.tail;
        if (parts > a.max_parts) parts = a.max_parts;
        if (parts > KS / 4) parts = KS / 4;                       // at least two k blocks per range
        if (parts >= 2) {
            r.sp = ((KS + parts - 1) / parts + 1) & ~1;           // whole k16 pairs
            r.parts = (KS + r.sp - 1) / r.sp;
        }
    }
    return r;
}

// unit u (local index inside XCD `xcd`): bins xcd, xcd + 8, ...; inside a bin the type-0 units (m pair major, n tile minor), then
// the type-1 units of the odd last panel
__host__ __device__ static inline X3RingUnit x3r_unit(const X3RingArgs& a, int xcd, int u) {
    X3RingUnit w;
    const int bi = u / a.upb, rem = u - bi * a.upb, n0 = a.pairs * a.NTt;
    w.bin = bi * 8 + xcd;
    if (rem < n0) {
        const int mp = rem / a.NTt, nt = rem - mp * a.NTt;
        w.type = 0; w.pa0 = 2 * mp; w.pa1 = 2 * mp + 1; w.pb0 = nt; w.pb1 = nt;
        w.row0 = mp * 256; w.col0 = nt * 128;
    } else {
        const int q = rem - n0;
        w.type = 1; w.pa0 = a.PM - 1; w.pa1 = a.PM - 1; w.pb0 = 2 * q; w.pb1 = 2 * q + 1 < a.NTt ? 2 * q + 1 : 2 * q;
        w.row0 = (a.PM - 1) * 128; w.col0 = q * 256;
    }
    return w;
}

// element e of a unit's register-order slab ([wave][tile i*2+j][r][lane]) -> (row, col) inside the unit
__host__ __device__ static inline void x3r_slab_rc(int type, int e, int& row, int& col) {
    const int lane = e & 63, r = (e >> 6) & 15, t = (e >> 10) & 3, wave = e >> 12;
    const int wm = wave >> 1, wn = wave & 1, i = t >> 1, j = t & 1;
    const int rr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), cc = wn * 64 + j * 32 + (lane & 31);
    if (type == 0) { row = wm * 64 + rr; col = cc; }
    else { row = (wm & 1) * 64 + rr; col = (wm >> 1) * 128 + cc; }
}

#if defined(__HIP_DEVICE_COMPILE__)
#define X3R_DEVICE_BODY 1
#else
#define X3R_DEVICE_BODY 0
#endif

template <int N> __device__ __forceinline__ void x3r_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((address_space(3))) void* x3r_lds_ptr;
typedef unsigned x3r_u4 __attribute__((ext_vector_type(4)));

// one item of a workgroup's list: a whole unit, or one K range of a split tail unit (all fields wave-uniform)
struct X3RingItem {
    X3RingUnit un;
    int ks0, nst, part;                      // first k16 sub-stage, sub-stages, K range index (-1: the whole unit)
    unsigned so0, so1, so2;                  // byte offsets of the three slices' panels (k block 0) in Ap / Bp
};
__device__ __forceinline__ X3RingItem x3r_item(const X3RingArgs& a, const X3RingXcd& xs, int xcd, int lw, int it) {
    X3RingItem m;
    const int KS = a.K / 16;
    int u;
    if (it < xs.rounds) { u = it * a.wgx + lw; m.ks0 = 0; m.nst = KS; m.part = -1; }
    else if (xs.parts > 1) {
        u = xs.rounds * a.wgx + lw / xs.parts; m.part = lw % xs.parts;
        m.ks0 = m.part * xs.sp; m.nst = (m.ks0 + xs.sp < KS ? m.ks0 + xs.sp : KS) - m.ks0;
    } else { u = xs.rounds * a.wgx + lw; m.ks0 = 0; m.nst = KS; m.part = -1; }
    m.un = x3r_unit(a, xcd, u);
    const unsigned pa = (unsigned)(a.K / 32) * X3R_PANEL_STAGE;
    m.so0 = ((unsigned)m.un.bin * a.PM + m.un.pa0) * pa;
    m.so1 = m.un.type ? ((unsigned)m.un.bin * a.NTt + m.un.pb0) * pa : ((unsigned)m.un.bin * a.PM + m.un.pa1) * pa;
    m.so2 = ((unsigned)m.un.bin * a.NTt + m.un.pb1) * pa;
    return m;
}

// KNOBS (measurement; 0 in production): bit 0 no MFMAs, bit 1 no DMA after the first unit's prologue, bit 2 no epilogue stores, bit 3 every DMA
// reads k block 0 (cache-resident operands), bit 4 clock report
template <int KNOBS>
__global__ __launch_bounds__(512, 2) void gemm_x3_ring_kernel(const X3RingArgs a) {
#if X3R_DEVICE_BODY
    __shared__ __attribute__((aligned(16))) unsigned char sm[X3R_LDS];      // ONE LDS object beside the LDS-DMA instructions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const X3RingXcd xs = x3r_xcd(a, xcd);
    const int n_items = xs.rounds + ((xs.parts > 1 ? lw < xs.tail * xs.parts : lw < xs.tail) ? 1 : 0);
    if (n_items == 0) return;
    // knob bit 4: workgroup 0 reports (shader-clock cycles, 100 MHz ticks) of its run at the end of the slab buffer: their ratio
    // is the clock the kernel actually ran at
    unsigned long long clk0 = 0, rt0 = 0;
    if (KNOBS & 16) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    // DMA: per slice 12 wave-instructions of 1 KB (plane = J >> 2, rows (J & 3) * 32 + lane / 2, half = lane & 1); wave w issues
    // J = w and, for w < 4, J = w + 8.  voff: byte offset inside a (panel, k block) stage for sub-stage 0; sub-stage 1 = ^ 32.
    const bool two = wave < 4;
    int voff0, voff1;
    {
        const int J0 = wave, row0 = (J0 & 3) * 32 + (lane >> 1);
        voff0 = (J0 >> 2) * 8192 + row0 * 64 + ((((lane & 1) ^ ((row0 >> 3) & 1)) ^ x3_sw(row0)) << 4);
        const int J1 = (wave & 3) + 8, row1 = (J1 & 3) * 32 + (lane >> 1);
        voff1 = (J1 >> 2) * 8192 + row1 * 64 + ((((lane & 1) ^ ((row1 >> 3) & 1)) ^ x3_sw(row1)) << 4);
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.Ap), 0, (int)a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.Bp), 0, (int)a.b_bytes, 0x00020000);

    // sub-stage t (relative to the item's first) -> ring slot t & 3
    auto dma_stage = [&](const X3RingItem& m, int t) {
        const int ks = m.ks0 + t;
        const unsigned kbo = (KNOBS & 8) ? 0u : (unsigned)(ks >> 1) * X3R_PANEL_STAGE;     // (knob: every stage re-reads k block 0)
        const int sx = (ks & 1) << 5;
        unsigned char* dst = sm + (t & 3) * X3R_SLOT + wave * 1024;
        const int v0 = voff0 ^ sx, v1 = voff1 ^ sx;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (x3r_lds_ptr)dst, 16, v0, (int)(m.so0 + kbo), 0, 0);
        if (m.un.type) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (x3r_lds_ptr)(dst + X3R_SLICE), 16, v0, (int)(m.so1 + kbo), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (x3r_lds_ptr)(dst + X3R_SLICE), 16, v0, (int)(m.so1 + kbo), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (x3r_lds_ptr)(dst + 2 * X3R_SLICE), 16, v0, (int)(m.so2 + kbo), 0, 0);
        if (two) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (x3r_lds_ptr)(dst + 8192), 16, v1, (int)(m.so0 + kbo), 0, 0);
            if (m.un.type) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (x3r_lds_ptr)(dst + X3R_SLICE + 8192), 16, v1, (int)(m.so1 + kbo), 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (x3r_lds_ptr)(dst + X3R_SLICE + 8192), 16, v1, (int)(m.so1 + kbo), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (x3r_lds_ptr)(dst + 2 * X3R_SLICE + 8192), 16, v1, (int)(m.so2 + kbo), 0, 0);
        }
    };
    // counted wait: everything but `younger` DMA sub-stages (3 or 6 instructions each) and `stores` epilogue stores has landed
    auto wait_for = [&](bool younger, bool stores) {
        if (two) {
            if (stores) { if (younger) x3r_wait_vm<6 + X3R_STORES>(); else x3r_wait_vm<X3R_STORES>(); }
            else { if (younger) x3r_wait_vm<6>(); else x3r_wait_vm<0>(); }
        } else {
            if (stores) { if (younger) x3r_wait_vm<3 + X3R_STORES>(); else x3r_wait_vm<X3R_STORES>(); }
            else { if (younger) x3r_wait_vm<3>(); else x3r_wait_vm<0>(); }
        }
    };
    // opens sub-stage t: sub-stage t + 1 has landed for every wave after the barrier; then sub-stage t + 3 is requested.
    // st: the previous unit's X3R_STORES epilogue stores (issued after this unit's first three requests) may still be in flight
    // (t = 0: they are younger than sub-stage 2's request; t = 1: younger than sub-stage 2's, older than sub-stage 3's)
    auto top = [&](const X3RingItem& m, int t, bool st) {
        if (t + 1 < m.nst) wait_for(t + 2 < m.nst, st);
        __builtin_amdgcn_s_barrier();
        if (t + 3 < m.nst && !(KNOBS & 2)) dma_stage(m, t + 3);
    };

    const int lane_addr = r32 * 32 + ((h ^ ((r32 >> 3) & 1)) << 4);
    f32x16 acc[2][2], cor[2][2];
    x3_bf16x8 A0[3], A1[3], X[3], Y[3];
    // three fragments (planes) of row tile `rt` of the wave's A (or B) rows from sub-stage t's slot
    auto ld3 = [&](x3_bf16x8 (&d)[3], int t, int base, int rt) {
        const unsigned char* s = sm + (t & 3) * X3R_SLOT + base + rt * 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) d[p] = __builtin_bit_cast(x3_bf16x8, *reinterpret_cast<const uint4*>(s + p * 4096));
    };
    // the 6 MFMAs of one 32 x 32 output tile: five correction products into `c`, the leading one into `m` (x3_stage_mfma's order)
    auto quarter = [&](const x3_bf16x8 (&A)[3], const x3_bf16x8 (&B)[3], f32x16& m, f32x16& c) {
        if (KNOBS & 1) return;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], c, 0, 0, 0);
        m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], m, 0, 0, 0);
    };

    X3RingItem cur = x3r_item(a, xs, xcd, lw, 0);
    {
        const int np = cur.nst < 3 ? cur.nst : 3;
        for (int t = 0; t < np; ++t) dma_stage(cur, t);
    }
    for (int it = 0; it < n_items; ++it) {
        // (item `cur`: its first three sub-stages are in flight)
        int a_addr, b_addr;
        if (cur.un.type == 0) { a_addr = (wm >> 1) * X3R_SLICE + (wm & 1) * 2048 + lane_addr; b_addr = 2 * X3R_SLICE + wn * 2048 + lane_addr; }
        else { a_addr = (wm & 1) * 2048 + lane_addr; b_addr = (1 + (wm >> 1)) * X3R_SLICE + wn * 2048 + lane_addr; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }
        const bool st_prev = it > 0;
        const int nst = cur.nst;
        top(cur, 0, st_prev);
        ld3(A0, 0, a_addr, 0); ld3(X, 0, b_addr, 0);
        // The four 32 x 32 tiles of a sub-stage in the order (0,0) (0,1) (1,1) (1,0): one operand changes per step, and it is
        // read while the previous step's MFMAs run (the column-0 fragments are read twice per sub-stage: four fragment sets
        // live instead of seven).  The B registers exchange roles every sub-stage (X / Y), hence two sub-stages per iteration.
        for (int t = 0; t < nst; t += 2) {
            ld3(Y, t, b_addr, 1);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A0, X, acc[0][0], cor[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            ld3(A1, t, a_addr, 1);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A0, Y, acc[0][1], cor[0][1]);
            __builtin_amdgcn_sched_barrier(0);
            ld3(X, t, b_addr, 0);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A1, Y, acc[1][1], cor[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            ld3(A0, t + 1, a_addr, 0); ld3(Y, t + 1, b_addr, 0);        // sub-stage t + 1 landed before the barrier of t
            __builtin_amdgcn_sched_barrier(0);
            quarter(A1, X, acc[1][0], cor[1][0]);
            __builtin_amdgcn_sched_barrier(0);
            top(cur, t + 1, st_prev && t == 0);
            ld3(X, t + 1, b_addr, 1);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A0, Y, acc[0][0], cor[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            ld3(A1, t + 1, a_addr, 1);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A0, X, acc[0][1], cor[0][1]);
            __builtin_amdgcn_sched_barrier(0);
            ld3(Y, t + 1, b_addr, 0);
            __builtin_amdgcn_sched_barrier(0);
            quarter(A1, X, acc[1][1], cor[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nst) { ld3(A0, t + 2, a_addr, 0); ld3(X, t + 2, b_addr, 0); }
            __builtin_amdgcn_sched_barrier(0);
            quarter(A1, Y, acc[1][0], cor[1][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nst) top(cur, t + 2, false);
        }
        __builtin_amdgcn_s_barrier();                       // every wave has read its last fragments: the slots are free
        // this item's outputs stay in registers while the next item's first sub-stages are requested
        X3RingItem nxt = cur;
        if (it + 1 < n_items) {
            nxt = x3r_item(a, xs, xcd, lw, it + 1);
            if (!(KNOBS & 2)) {
                const int np = nxt.nst < 3 ? nxt.nst : 3;
                for (int t = 0; t < np; ++t) dma_stage(nxt, t);
            }
        }
        if (cur.part >= 0) {
            // K range of a split tail unit: raw partial sums in register order (coalesced 256-byte rows); last item of this workgroup
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));                // (keeps the 64 store addresses out of the loops' preheader: they spilled)
            float* sl = a.slabs + ((size_t)(xcd * a.wgx + lw)) * X3R_UNIT_FLOATS + wave * 4096 + lane_e;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sl[((i * 2 + j) * 16 + r) * 64] = acc[i][j][r] + cor[i][j][r];
        } else {
            // epilogue: per 16 x 32 piece 8 ds_write_b32 -> 2 ds_read_b128 -> 2 buffer stores of 16 B (8 lanes = one 128-byte row piece)
            float* scr = reinterpret_cast<float*>(sm + X3R_SCR + wave * 2048);
            const X3RingUnit& un = cur.un;
            const int row_w = un.type == 0 ? un.row0 + wm * 64 : un.row0 + (wm & 1) * 64;
            const int col_w = un.type == 0 ? un.col0 + wn * 64 : un.col0 + (wm >> 1) * 128 + wn * 64;
            const bool col_ok = col_w < a.N && !(un.type == 1 && (wm >> 1) == 1 && un.pb1 == un.pb0);
            float* Cb = a.C + (size_t)un.bin * a.M * a.N;
            const unsigned long long cbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)Cb >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)Cb);
            const int clen = __builtin_amdgcn_readfirstlane((KNOBS & 4) || !col_ok ? 0 : (int)((unsigned)a.M * (unsigned)a.N * 4u));
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(cbu), 0, clen, 0x00020000);
            const int rd_row = lane >> 3, rd_c4 = (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int r = hf * 8 + q, lr = (q & 3) + 8 * (q >> 2) + 4 * h;
                            scr[lr * 32 + r32] = acc[i][j][r] + cor[i][j][r];
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(&scr[(q * 8 + rd_row) * 32 + rd_c4]);
                            const unsigned row = (unsigned)(row_w + i * 32 + hf * 16 + q * 8 + rd_row);
                            const unsigned off = (row * (unsigned)a.N + (unsigned)(col_w + j * 32 + rd_c4)) * 4u;   // rows >= M: past the descriptor
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(x3r_u4, v), rsC, (int)off, 0, 0);
                        }
                    }
        }
        cur = nxt;
    }
    if ((KNOBS & 16) && blockIdx.x == 0 && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.slabs + (size_t)8 * a.wgx * X3R_UNIT_FLOATS);
        o[0] = __builtin_readcyclecounter() - clk0; o[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

// sums the K ranges of the split tail units in slab order and writes C (element = one output; slabs are in register order)
__global__ __launch_bounds__(256) void x3r_combine_kernel(const X3RingArgs a) {
    const int xcd = blockIdx.y;
    const X3RingXcd xs = x3r_xcd(a, xcd);
    if (xs.parts < 2) return;
    const int tu = blockIdx.x / (X3R_UNIT_FLOATS / 1024), e0 = (blockIdx.x % (X3R_UNIT_FLOATS / 1024)) * 1024 + threadIdx.x;
    if (tu >= xs.tail) return;
    const X3RingUnit un = x3r_unit(a, xcd, xs.rounds * a.wgx + tu);
    const float* sl = a.slabs + ((size_t)(xcd * a.wgx + tu * xs.parts)) * X3R_UNIT_FLOATS;
    float* Cb = a.C + (size_t)un.bin * a.M * a.N;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int e = e0 + v * 256;
        float s = sl[e];
        for (int p = 1; p < xs.parts; ++p) s += sl[(size_t)p * X3R_UNIT_FLOATS + e];
        int row, col;
        x3r_slab_rc(un.type, e, row, col);
        row += un.row0; col += un.col0;
        if (row < a.M && col < a.N && !(un.type == 1 && col >= un.col0 + 128 && un.pb1 == un.pb0)) Cb[(size_t)row * a.N + col] = s;
    }
}
