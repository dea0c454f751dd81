// Potts expert on gfx950: Delta-H and dH/dx for a batch of chains in ONE pass over the couplings.
//
// Replaces PottsModel.hamiltonian/forward (reference ppde/nets.py:282-299: two einsums + bias) and the
// autograd of it (ppde/energy.py:108). With M = (J + J^T)/2 over flattened (residue, letter) pairs and a
// one-hot state x,  S = M x  is a GATHER:  S[(i,k)] = sum_j M[(i,k),(j,a_j)], and
//     grad[(i,k)] = h[(i,k)] + S[(i,k)],       H = sum_i ( h[(i,a_i)] + S[(i,a_i)] / 2 ).
//
// Work split: one workgroup owns 4 output columns (i, k0..k0+3) for ALL chains of its chain block, so
// every coupling byte is fetched from HBM exactly once per launch; the 4-column slab of M
// ([Lw*20 rows][4] floats, 25.6 KB for L'=80) is staged in LDS and each chain's 80 rows are then gathered
// with ds_read_b128. Lanes are chains, waves are the four residue PARTS: lanes of a wave read the same
// residue j at the same time, so chains that agree at j (populations near the wild type) hit the same
// LDS row and broadcast.
//
// Summation order (fixed, independent of batch size / launch shape, so sharding chains over GPUs cannot
// change a bit): the padded window is 4 parts x NC chunks x 4 residues; each part is summed in residue
// order, then parts combine as (p0 + p1) + (p2 + p3).
#pragma once
#include "common.h"

// Tiled coupling layout: Jt[tile][part][chunk][q][letter][c], tile = i*5 + k0/4, residue
// j = part*4*NC + chunk*4 + q (rows of residues j >= Lp are zero). One (tile, part) region is
// NC*80 rows of 16 B, contiguous, and is streamed by exactly one wavefront.
__device__ __host__ __forceinline__ size_t jt_tile_float4s(int NC) { return (size_t)NC * 320; }

// Build Jt from the reference-layout couplings J[i][j][k][l] (symmetrising on the fly).
__global__ void potts_prepare_kernel(const float* __restrict__ J, float* __restrict__ Jt, int Lp, int NC) {
    const int Lw = 16 * NC;
    const size_t total = (size_t)Lp * 5 * Lw * 20 * 4;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        int c = o & 3;
        size_t r = o >> 2;
        int l = r % 20; r /= 20;
        int q = r & 3; r >>= 2;
        int chunk = r % NC; r /= NC;
        int part = r & 3; r >>= 2;
        int tile = (int)r;
        int i = tile / 5, k = (tile % 5) * 4 + c;
        int j = part * 4 * NC + chunk * 4 + q;
        float v = 0.f;
        if (j < Lp) {
            float a = J[(((size_t)i * Lp + j) * 20 + k) * 20 + l];
            float b = J[(((size_t)j * Lp + i) * 20 + l) * 20 + k];
            v = 0.5f * (a + b);
        }
        Jt[o] = v;
    }
}

struct PottsArgs {
    const float4* Jt;        // tiled symmetrised couplings
    const float* h;          // [Lp*20]
    const uint8_t* idx;      // states [n][Ls]
    float* grad;             // [slots][n][N]   (slot stride = n*N)
    float* epart;            // [slots][n][Lp]  per-residue energy terms h + S/2 at the chain's letter
    int slot;                // which evaluation slot to write (0 = current state, 1 = proposal)
    int n;                   // chains in the buffers (slot stride)
    int b_off, n_sub;        // this launch covers chains [b_off, b_off + n_sub)
    unsigned long long* dbg; // stamp buffer (diagnostic build)
    Geom g;
};

// s_waitcnt vmcnt(n) with a run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define PPDE_W(k) case k: __builtin_amdgcn_s_waitcnt(((k) & 0xF) | (((k) >> 4) << 14) | 0x0F70); break;
    switch (n) {
        PPDE_W(0) PPDE_W(1) PPDE_W(2) PPDE_W(3) PPDE_W(4) PPDE_W(5) PPDE_W(6) PPDE_W(7)
        PPDE_W(8) PPDE_W(9) PPDE_W(10) PPDE_W(11) PPDE_W(12) PPDE_W(13) PPDE_W(14) PPDE_W(15)
        PPDE_W(16) PPDE_W(17) PPDE_W(18) PPDE_W(19) PPDE_W(20) PPDE_W(21) PPDE_W(22) PPDE_W(23)
        PPDE_W(24) PPDE_W(25) PPDE_W(26) PPDE_W(27) PPDE_W(28) PPDE_W(29) PPDE_W(30) PPDE_W(31)
        default: break;   // more than 31 younger operations outstanding: nothing to wait for yet
    }
#undef PPDE_W
}

// LDS-DMA issued from inline asm is invisible to hipcc's wait-count bookkeeping, so the counted vmcnt waits
// below are the only ones in the gather loop (with __builtin_amdgcn_global_load_lds hipcc drains vmcnt(0) before
// the first ds_read of every chunk). M0 carries the wave-uniform LDS base and is restored (cdna_hip_programming.md §5.7).
// (gfx9 LDS instructions do not read M0, so it is not restored: nothing else in these kernels uses it.)
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
    return (uint32_t)(uintptr_t)p;   // low 32 bits of a generic LDS address = LDS offset (callers keep it wave-uniform)
}

// LDS plan of one workgroup: [4 parts][NC*80] float4 slab | [4][CPB] float4 part sums | raw state rows [CPB][Ls] bytes
__host__ __device__ inline size_t potts_lds_bytes(int NC, int NG, int Ls) {
    return ((size_t)4 * NC * 80 + (size_t)4 * NG * 64) * 16 + (((size_t)NG * 64 * Ls + 1023) & ~(size_t)1023) + 1024;
}
// Long windows (NC > POTTS_RING_CHUNKS): each wave streams its slab rows through a ring of 8 chunks = 10 pieces
// (10 KiB) instead of holding them all, which keeps two workgroups per CU (GFP, L' = 237: 113 KB -> 73 KB) and the
// DMA queue fed while the wave gathers. The part sums are exchanged through the (then free) rings.
#define POTTS_RING_CHUNKS 8
#define POTTS_RING_PIECES 10
__host__ __device__ inline size_t potts_ring_lds_bytes(int NG, int Ls) {
    return (size_t)4 * POTTS_RING_CHUNKS * 80 * 16 + (((size_t)NG * 64 * Ls + 1023) & ~(size_t)1023) + 1024;
}

// Body of one workgroup: `tile` = 4 output columns, `by` = block of NG*64 chains. A __device__ function so that
// the kernel below and the fused experts launch (ppde_api.hip: k_experts) share it.
template <int NG, bool RING = false>   // NG groups of 64 chains per workgroup
__device__ __forceinline__ void potts_body(const PottsArgs& a, const int tile, const int by, float4* smem) {
    static_assert(!RING || NG <= 2, "the part sums of a ring workgroup live in one wave's 10 KiB ring");
    const Geom g = a.g;
    const int NC = g.NC;
    constexpr int CPB = NG * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every address derived from it stays in SGPRs
    const int b0 = a.b_off + by * CPB;
    const int b_end = a.b_off + a.n_sub;
    [[maybe_unused]] const bool stamp = tile == 0 && by == 0;
    PPDE_STAMP(a.dbg, 0, stamp);
    const int slab_rows = RING ? POTTS_RING_CHUNKS * 80 : NC * 80;   // float4 rows of LDS per wave
    float4* sT = smem + (size_t)part * slab_rows;               // this wave's rows of the slab (or its ring)
    float4* sR = RING ? smem : smem + (size_t)4 * slab_rows;    // [4][CPB] part sums (ring: wave p's sums in wave p's ring)
    const int sr_stride = RING ? slab_rows : CPB;
    uint8_t* sS = (uint8_t*)(RING ? smem + (size_t)4 * slab_rows : sR + 4 * CPB);   // raw state rows of this chain block [CPB][Ls]

    // ---- LDS-DMA, 1 KiB a piece: first the chain block's state rows (one contiguous, coalesced range shared by
    //      the four waves), then this wave's own slab rows. Ls/4 is odd, so the strided letter reads below are
    //      bank-conflict free. (Plain register staging + ds_write in this structure measured 8.7 us vs 5.0.)
    const int state_bytes = min(CPB, b_end - b0) * g.Ls;
    const char* ssrc = (const char*)(a.idx + (size_t)b0 * g.Ls);
    const int region_bytes = NC * 1280;
    const int npieces = (region_bytes + 1023) >> 10;
    const char* src = (const char*)(a.Jt + ((size_t)tile * 4 + part) * NC * 80);
    const int spieces = (state_bytes + 1023) >> 10;
    for (int p = part; p < spieces; p += 4) {
        const int off = p * 1024 + lane * 16;
        if (off < state_bytes) glds16_asm(ssrc + off, lds_offset_of(sS + p * 1024));
    }
    int issued = 0;                                              // slab pieces requested so far
    const int first = RING ? min(npieces, POTTS_RING_PIECES) : npieces;
    for (; issued < first; ++issued) {
        const int off = issued * 1024 + lane * 16;
        if (off < region_bytes) glds16_asm(src + off, lds_offset_of(sT + issued * 64));
    }
    PPDE_STAMP(a.dbg, 1, stamp);
    wait_vmcnt(issued);                                         // my state pieces have landed (issued first) ...
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // ... and so have the other waves'
    PPDE_STAMP(a.dbg, 2, stamp);
    const uint8_t* myrow = sS + (size_t)lane * g.Ls + g.sh + g.i0 + 4 * part * NC;

    // ---- gather: wave = part, lane = chain; chunk ck is ready once its pieces have landed
    float4 acc[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ck = 0; ck < NC; ++ck) {
        const int need = min(((ck + 1) * 1280 + 1023) >> 10, npieces);
        wait_vmcnt(issued - need);
        asm volatile("" ::: "memory");
        const float4* rows = sT + (RING ? ck % POTTS_RING_CHUNKS : ck) * 80;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const uint32_t w = *(const uint32_t*)(myrow + (size_t)gi * 64 * g.Ls + 4 * ck);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t letter = min((w >> (8 * q)) & 0xFFu, 19u);
                const float4 r = rows[q * 20 + letter];
                acc[gi].x += r.x; acc[gi].y += r.y; acc[gi].z += r.z; acc[gi].w += r.w;
            }
        }
        if (RING && (ck & 3) == 3 && issued < npieces) {        // chunks 4h .. 4h+3 = pieces 5h .. 5h+4 are consumed: refill them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (their LDS reads have returned)
            const int upto = min(issued + POTTS_RING_PIECES / 2, npieces);
            for (; issued < upto; ++issued) {
                const int off = issued * 1024 + lane * 16;
                if (off < region_bytes) glds16_asm(src + off, lds_offset_of(sT + (issued % POTTS_RING_PIECES) * 64));
            }
        }
    }
    PPDE_STAMP(a.dbg, 3, stamp);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) sR[part * sr_stride + gi * 64 + lane] = acc[gi];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    PPDE_STAMP(a.dbg, 4, stamp);
    // ---- combine parts, add fields, write gradient columns and the energy term
    const int i = tile / 5, k0 = (tile % 5) * 4;
    const float4 h4 = *(const float4*)(a.h + i * 20 + k0);
    for (int cl = tid; cl < CPB; cl += 256) {
        const int b = b0 + cl;
        if (b >= b_end) continue;
        const float4 s0 = sR[cl], s1 = sR[sr_stride + cl], s2 = sR[2 * sr_stride + cl], s3 = sR[3 * sr_stride + cl];
        float4 S;
        S.x = (s0.x + s1.x) + (s2.x + s3.x);
        S.y = (s0.y + s1.y) + (s2.y + s3.y);
        S.z = (s0.z + s1.z) + (s2.z + s3.z);
        S.w = (s0.w + s1.w) + (s2.w + s3.w);
        const int slot = a.slot;
        float4* out = (float4*)(a.grad + ((size_t)slot * a.n + b) * g.N + (g.i0 + i) * 20 + k0);
        *out = make_float4(S.x + h4.x, S.y + h4.y, S.z + h4.z, S.w + h4.w);
        const uint32_t letter = sS[(size_t)cl * g.Ls + g.sh + g.i0 + i];
        const int kk = (int)letter - k0;
        if (kk >= 0 && kk < 4) {
            const float sv = kk == 0 ? S.x : kk == 1 ? S.y : kk == 2 ? S.z : S.w;
            const float hv = kk == 0 ? h4.x : kk == 1 ? h4.y : kk == 2 ? h4.z : h4.w;
            a.epart[((size_t)slot * a.n + b) * g.Lp + i] = hv + 0.5f * sv;
        }
    }
    PPDE_STAMP(a.dbg, 5, stamp);
}

template <int NG, bool RING = false>
__global__ __launch_bounds__(256) void potts_energy_grad_kernel(PottsArgs a) {
    warm_kernargs<sizeof(PottsArgs)>();
    extern __shared__ float4 smem[];
    potts_body<NG, RING>(a, blockIdx.x, blockIdx.y, smem);
}

// H = sum_i epart[i] in a fixed tree with fp64 partials (one wave); returns the same value in all lanes.
__device__ __forceinline__ float potts_hamiltonian_from_parts(const float* ep, int Lp) {
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < Lp; i += 64) s += (double)ep[i];
    return (float)wave_sum_d(s);
}

// e[b] = (H(b) - wt_H) for the stateless API (one wave per chain).
__global__ void potts_energy_finalize_kernel(const float* __restrict__ epart, int Lp, float wt_H,
                                             float* __restrict__ e, int n) {
    int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n) return;
    float H = potts_hamiltonian_from_parts(epart + (size_t)b * Lp, Lp);
    if ((threadIdx.x & 63) == 0) e[b] = H - wt_H;
}
