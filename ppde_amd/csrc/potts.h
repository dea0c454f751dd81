// Potts expert on gfx950: Delta-H and dH/dx for a batch of chains in ONE pass over the couplings.
//
// Replaces PottsModel.hamiltonian/forward (reference ppde/nets.py:282-299: two einsums + bias) and the
// autograd of it (ppde/energy.py:108). With M = (J + J^T)/2 over flattened (residue, letter) pairs and a
// one-hot state x,  S = M x  is a GATHER:  S[(i,k)] = sum_j M[(i,k),(j,a_j)], and
//     grad[(i,k)] = h[(i,k)] + S[(i,k)],       H = sum_i ( h[(i,a_i)] + S[(i,a_i)] / 2 ).
//
// Work split: one workgroup owns 4 output columns (i, k0..k0+3) for ALL chains of its chain block, so
// every coupling byte is fetched from HBM exactly once per launch; the 4-column slab of M
// ([Lw*20 rows][4] floats, 25.6 KB for L'=80) is streamed into LDS by LDS-DMA and each chain's 80 rows are then
// gathered with ds_read_b128. Lanes are chains, waves are the four residue PARTS: lanes of a wave read the same
// residue j at the same time, so chains that agree at j (populations near the wild type) hit the same
// LDS row and broadcast. The chains' letters come from a transposed copy of the states (T4, below) with one
// coalesced 16-byte load per lane and four chunks: nothing but couplings goes through LDS.
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

// Transposed copy of the window letters, kept next to every state buffer ("T4" layout): for part p and group h of
// four chunks, T4[(p*G + h) * n_pad + b] is ONE 16-byte word of chain b holding the letters of chunks 4h .. 4h+3 of
// that part (chunk ck, residue q -> byte 4*(ck & 3) + q; residue j = 4*NC*p + 4*ck + q of the padded window).
// Chains are contiguous, so a wave (lane = chain) fetches the letters of four chunks with one coalesced
// global_load_dwordx4 straight into registers: no state rows in LDS, no barrier before the gather. Whoever writes a
// state row also writes its T4 bytes (state_t4_offset below; chain kernels, k_state_to_t4).
__host__ __device__ inline int potts_groups(int NC) { return (NC + 3) >> 2; }
__host__ __device__ inline int potts_t4_pad(int n) { return ((n + 63) & ~63) + 256; }   // lanes past the last chain stay in bounds
__host__ __device__ inline size_t potts_t4_words(int NC, int n_pad) { return (size_t)4 * potts_groups(NC) * n_pad * 4; }   // uint32s
// byte offset of window residue wl (0 <= wl < 16*NC) of chain b in a T4 buffer
__host__ __device__ inline size_t state_t4_offset(int NC, int n_pad, int b, int wl) {
    const int p = wl / (4 * NC), r = wl - p * 4 * NC, ck = r >> 2;
    return (((size_t)(p * potts_groups(NC) + (ck >> 2)) * n_pad + b) << 4) + ((ck & 3) << 2) + (r & 3);
}

struct PottsArgs {
    const float4* Jt;        // tiled symmetrised couplings
    const float* h;          // [Lp*20]
    const uint32_t* idxT;    // T4 copy of the states (see above)
    int n_pad;               // chains per T4 row
    float* grad;             // [slots][n][N]   (slot stride = n*N)
    float* epart;            // [slots][n][Lp]  per-residue energy terms h + S/2 at the chain's letter
    int slot;                // which evaluation slot to write (0 = current state, 1 = proposal)
    int n;                   // chains in the buffers (slot stride)
    int b_off, n_sub;        // this launch covers chains [b_off, b_off + n_sub)
    unsigned long long* dbg; // stamp buffer (diagnostic build)
    int dbg_wg_base;         // first per-workgroup record of this launch's Potts tiles (the fused experts launch keeps the CNN's in front)
    Geom g;
};

// s_waitcnt vmcnt(n) with a run-time n (the instruction takes an immediate). MAXW bounds the switch: a request for
// more than MAXW - 1 outstanding operations waits for MAXW - 1 (waiting for fewer is always safe), which keeps the
// expanded code of the eight call sites of the gather small.
template <int MAXW>
__device__ __forceinline__ void wait_vmcnt(int n) {
    static_assert(MAXW <= 32, "extend the switch");
#define PPDE_W(k) case k: __builtin_amdgcn_s_waitcnt(((k) & 0xF) | (((k) >> 4) << 14) | 0x0F70); break;
#define PPDE_W8(k) PPDE_W(k) PPDE_W(k + 1) PPDE_W(k + 2) PPDE_W(k + 3) PPDE_W(k + 4) PPDE_W(k + 5) PPDE_W(k + 6) PPDE_W(k + 7)
    switch (n < 0 ? 0 : (n >= MAXW ? MAXW - 1 : n)) {
        PPDE_W8(0) PPDE_W8(8) PPDE_W8(16) PPDE_W8(24)
        default: break;
    }
#undef PPDE_W8
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
// Register loads hipcc must not count either (cdna_hip_programming.md §5.7 item 1, form ii): a load hipcc sees would be
// waited for with the count of ITS loads only, i.e. behind every LDS-DMA piece issued after it. The destination is
// not valid until the caller's own counted wait; landed() then pins the first use behind that wait.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 gload16_asm(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t gload_ubyte_asm(const void* p) {
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void landed(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void landed(uint32_t& v) { asm volatile("" : "+v"(v)); }

// Workgroups with equal (block id mod 8) share an XCD and its L2 (observed dispatch order; speed only, any placement
// is correct). Give each of the eight groups a CONTIGUOUS range of work items: the five column tiles of a residue
// and the chain blocks of a tile then meet in one L2, which merges their 16-byte gradient pieces into whole lines
// before they leave for memory (they used to reach the fabric as 2.1x the payload in partial-line writes) and
// serves a tile's couplings to its second chain block.
__device__ __forceinline__ int xcd_contiguous(int w, int total) {
    const int q = total >> 3, r = total & 7, x = w & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
}

__device__ __forceinline__ void glds4_asm(const void* gsrc, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}

// LDS plan of one workgroup: [4 parts][NC*80] float4 slab | [4][CPB] float4 part sums
__host__ __device__ inline size_t potts_lds_bytes(int NC, int NG) {
    return ((size_t)4 * NC * 80 + (size_t)4 * NG * 64) * 16;
}
// Long windows (NC > 8): each wave streams its slab rows through a ring of POTTS_RING_CHUNKS chunks instead of
// holding them all. A chunk (4 residues x 20 letters x 16 B = 1280 B) is ONE 1-KiB and ONE 256-byte LDS-DMA
// instruction, so the ring is refilled chunk by chunk right behind the gather. Six chunks per wave = 30 KB per
// workgroup: five workgroups per CU, i.e. every tile of a GFP-sized window (1185) is resident at once and all of
// them stream from the first microsecond (with 73 KB per workgroup two fitted, with 40 KB four: the tiles beyond
// the resident set started 8-10 us late into a 18 us launch). The part sums are exchanged through the (then free) rings.
#define POTTS_RING_CHUNKS 6
__host__ __device__ inline size_t potts_ring_lds_bytes() { return (size_t)4 * POTTS_RING_CHUNKS * 80 * 16; }

// the part of the window residue wl lies in, without a division by the run-time chunk count (wl < 16*NC)
__device__ __forceinline__ int potts_part_of(int wl, int NC) { return (wl >= 4 * NC) + (wl >= 8 * NC) + (wl >= 12 * NC); }
__device__ __forceinline__ size_t state_t4_offset_dev(int NC, int n_pad, int b, int wl) {
    const int p = potts_part_of(wl, NC), r = wl - p * 4 * NC, ck = r >> 2;
    return (((size_t)(p * potts_groups(NC) + (ck >> 2)) * n_pad + b) << 4) + ((ck & 3) << 2) + (r & 3);
}

// Body of one workgroup: `tile` = 4 output columns, `by` = block of NG*64 chains. A __device__ function so that
// the kernel below and the fused experts launch (ppde_api.hip: k_experts) share it.
// GM (ring variant) = chunk groups whose letters a wave keeps in registers: the window may have up to 16*GM*... 4*GM chunks per part.
// NCC: the window's chunk count as a compile-time constant (0 = read g.NC at run time). With it every trip count, every DMA
// piece count and every counted s_waitcnt of the body is an immediate (the run-time form goes through wait_vmcnt's switch);
// the host instantiates NCC = 5 (windows of 65..80 residues: PABP_YEAST 80, UBE4B_MOUSE 76).
template <int NG, bool RING = false, int GM = 2, int NCC = 0>   // NG groups of 64 chains per workgroup
__device__ __forceinline__ void potts_body(const PottsArgs& a, const int tile, const int by, float4* smem) {
    static_assert(!RING || NG <= 2, "the part sums of a ring workgroup live in one wave's ring");
    static_assert(RING || GM == 2, "the resident slab holds at most 8 chunks per part");
    const Geom g = a.g;
    const int NC = NCC ? NCC : g.NC, G = potts_groups(NC);
    constexpr int CPB = NG * 64;
    static_assert(CPB <= 256, "one thread per chain in the epilogue");
    const int tid = threadIdx.x, lane = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every address derived from it stays in SGPRs
    const int b0 = a.b_off + by * CPB;
    const int b_end = a.b_off + a.n_sub;
    [[maybe_unused]] const bool stamp = tile == 0 && by == 0;
    [[maybe_unused]] const int wg_lin = a.dbg_wg_base + tile * ((a.n_sub + CPB - 1) / CPB) + by;
    PPDE_STAMP(a.dbg, 0, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 0);
    const int slab_rows = RING ? POTTS_RING_CHUNKS * 80 : NC * 80;   // float4 rows of LDS per wave
    float4* sT = smem + (size_t)part * slab_rows;               // this wave's rows of the slab (or its ring)
    float4* sR = RING ? smem : smem + (size_t)4 * slab_rows;    // [4][CPB] part sums (ring: wave p's sums in wave p's ring)
    const int sr_stride = RING ? slab_rows : CPB;

    // ---- the couplings first: nothing else delays the first LDS-DMA (resident: all 1-KiB pieces of the wave's region;
    //      ring: the first POTTS_RING_CHUNKS chunks, two instructions each)
    const int region_bytes = NC * 1280;
    const int npieces = (region_bytes + 1023) >> 10;
    const char* src = (const char*)(a.Jt + ((size_t)tile * 4 + part) * NC * 80);
    int issued = 0;                                              // resident: pieces requested; ring: chunks requested
    auto issue_chunk = [&](int c) {                              // ring: chunk c -> slot c mod R
        const char* sc = src + (size_t)c * 1280;
        float4* dst = sT + (c % POTTS_RING_CHUNKS) * 80;
        glds16_asm(sc + lane * 16, lds_offset_of(dst));
        glds4_asm(sc + 1024 + lane * 4, lds_offset_of(dst + 64));
    };
    auto issue_piece = [&](int p) {                              // resident: 1-KiB piece p of the wave's region
        const int off = p * 1024 + lane * 16;
        if (off < region_bytes) glds16_asm(src + off, lds_offset_of(sT + p * 64));
    };
    // chunk 0's DMAs open the wave's queue ...
    const int head = RING ? 2 : min(2, npieces);                 // DMA instructions ahead of the letters
    if constexpr (RING) { issue_chunk(0); issued = 1; }
    else { for (; issued < head; ++issued) issue_piece(issued); }
    // ---- ... then the chains' letters: at the tile's own residue (one byte per thread, for the energy term) and of
    //      this wave's chunk groups (one 16-byte word per lane, chain group and chunk group) ...
    const int i = tile / 5, k0 = (tile % 5) * 4;
    const uint8_t* T8 = (const uint8_t*)a.idxT;
    uint32_t my_letter = gload_ubyte_asm(T8 + state_t4_offset_dev(NC, a.n_pad, b0 + min(tid, CPB - 1), i));
    const u32x4* tw = (const u32x4*)a.idxT + (size_t)(part * G) * a.n_pad + b0 + lane;
    u32x4 w[GM][NG];
#pragma unroll
    for (int h = 0; h < GM; ++h) {
        if (RING && h >= G) break;                               // (resident: always two groups, the second clamped)
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) w[h][gi] = gload16_asm(tw + (size_t)(RING ? h : min(h, G - 1)) * a.n_pad + gi * 64);
    }
    const int nwords = (RING ? G : 2) * NG;
    // ---- ... then the rest of the first fill (resident: every remaining piece; ring: chunks 1 .. R-1)
    if constexpr (RING) { for (; issued < min(NC, POTTS_RING_CHUNKS); ++issued) issue_chunk(issued); }
    else { for (; issued < npieces; ++issued) issue_piece(issued); }
    // The wave's vector-memory queue, oldest first: [head DMAs] [letter byte] [letter words] [further DMAs ...]. All
    // waits are counted vmcnt on this order: entry k has landed once at most (issued_total - k - 1) younger ones
    // are outstanding. DMA instruction d (in DMA order) is entry d if d < head, else d + 1 + nwords.
    int issued_total = (RING ? 2 * issued : issued) + 1 + nwords;
    const float4 h4 = *(const float4*)(a.h + i * 20 + k0);      // (consumed in the epilogue)
    PPDE_STAMP(a.dbg, 1, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 1);

    // ---- gather: wave = part, lane = chain
    float4 acc[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < GM; ++h) {
        if (h >= G) break;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ck = 4 * h + c;
            if (ck >= NC) break;
            const int last_dma = RING ? 2 * ck + 1 : min(((ck + 1) * 1280 + 1023) >> 10, npieces) - 1;   // last DMA chunk ck needs
            int last = last_dma < head ? last_dma : last_dma + 1 + nwords;
            if (ck == 0) last = head + nwords;                   // (and the letters, which follow chunk 0's DMAs)
            wait_vmcnt<24>(issued_total - last - 1);
            asm volatile("" ::: "memory");
            if (ck == 0) {
#pragma unroll
                for (int hh = 0; hh < GM; ++hh) {
#pragma unroll
                    for (int gi = 0; gi < NG; ++gi) landed(w[hh][gi]);
                }
                landed(my_letter);
            }
            const float4* rows = sT + (RING ? ck % POTTS_RING_CHUNKS : ck) * 80;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const uint32_t wv = w[h][gi][c];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t letter = min((wv >> (8 * q)) & 0xFFu, 19u);
                    const float4 r = rows[q * 20 + letter];
                    acc[gi].x += r.x; acc[gi].y += r.y; acc[gi].z += r.z; acc[gi].w += r.w;
                }
            }
            if (RING && issued < NC) {                           // the slot just gathered takes chunk ck + R
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (its LDS reads have returned)
                issue_chunk(issued);
                ++issued;
                issued_total += 2;
            }
        }
    }
    PPDE_STAMP(a.dbg, 3, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (ring: nothing may still land in the rings)
    if (RING) __syncthreads();                                   // every wave is done reading its ring before sums overwrite it
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) sR[part * sr_stride + gi * 64 + lane] = acc[gi];
    __syncthreads();

    PPDE_STAMP(a.dbg, 4, stamp);
    // ---- combine parts, add fields, write gradient columns and the energy term: thread = chain
    const int cl = tid, b = b0 + cl;
    if (cl < CPB && b < b_end) {
        const float4 s0 = sR[cl], s1 = sR[sr_stride + cl], s2 = sR[2 * sr_stride + cl], s3 = sR[3 * sr_stride + cl];
        float4 S;
        S.x = (s0.x + s1.x) + (s2.x + s3.x);
        S.y = (s0.y + s1.y) + (s2.y + s3.y);
        S.z = (s0.z + s1.z) + (s2.z + s3.z);
        S.w = (s0.w + s1.w) + (s2.w + s3.w);
        const int slot = a.slot;
        float4* out = (float4*)(a.grad + ((size_t)slot * a.n + b) * g.N + (g.i0 + i) * 20 + k0);
        *out = make_float4(S.x + h4.x, S.y + h4.y, S.z + h4.z, S.w + h4.w);
        const int kk = (int)my_letter - k0;
        if (kk >= 0 && kk < 4) {
            const float sv = kk == 0 ? S.x : kk == 1 ? S.y : kk == 2 ? S.z : S.w;
            const float hv = kk == 0 ? h4.x : kk == 1 ? h4.y : kk == 2 ? h4.z : h4.w;
            a.epart[((size_t)slot * a.n + b) * g.Lp + i] = hv + 0.5f * sv;
        }
    }
    PPDE_STAMP(a.dbg, 5, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 3);
}

// grid = (tiles x chain blocks) workgroups in one dimension; consecutive work items (chain blocks of a tile, then the
// next tile) go to one XCD (xcd_contiguous)
template <int NG, bool RING = false, int GM = 2, int NCC = 0>
__global__ __launch_bounds__(256, RING ? (GM <= 4 ? 5 : 4) : 2) void potts_energy_grad_kernel(PottsArgs a, int nby) {
    warm_kernargs<sizeof(PottsArgs) + 8>();
    extern __shared__ float4 smem[];
    const int v = xcd_contiguous(blockIdx.x, gridDim.x);
    if (nby == 1) potts_body<NG, RING, GM, NCC>(a, v, 0, smem);
    else potts_body<NG, RING, GM, NCC>(a, v / nby, v % nby, smem);
}

// State rows [n][Ls] -> their T4 copy (API edge and initialisation; the chain kernels write both forms themselves).
// One thread per (chain, window residue slot).
__global__ void k_state_to_t4(const uint8_t* __restrict__ rows, uint32_t* __restrict__ T, int n, int n_pad, Geom g) {
    const int W = 16 * g.NC;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * W) return;
    const int b = t / W, wl = t - b * W, l = g.i0 + wl;
    ((uint8_t*)T)[state_t4_offset(g.NC, n_pad, b, wl)] = l < g.L ? rows[(size_t)b * g.Ls + g.sh + l] : 0;
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
