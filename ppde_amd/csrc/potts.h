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

// Tiled coupling layout: Jt[tile][chunk][part][q][letter][c], tile = i*5 + k0/4, residue
// j = part*4*NC + chunk*4 + q (rows of residues j >= Lp are zero).
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
        int part = r & 3; r >>= 2;
        int chunk = r % NC; r /= NC;
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
    const uint8_t* cursel;   // per chain: slot holding the CURRENT gradient (NULL -> slot 0 is written)
    int slot_mode;           // 0: write slot `slot_fixed`; 1: write the slot not named by cursel[b]
    int slot_fixed;
    int n;
    int accumulate;          // 1: grad += (window columns already hold lamda * d fit/dx)
    Geom g;
};

template <int NG>   // NG groups of 64 chains per workgroup
__global__ __launch_bounds__(256) void potts_energy_grad_kernel(PottsArgs a) {
    extern __shared__ float4 smem[];
    const Geom g = a.g;
    const int NC = g.NC;
    const int CPB = NG * 64;
    float4* sT = smem;                                   // [NC*320] slab of M
    const int region0 = max(NC * 320, 4 * CPB);          // slab, later reused for the part sums
    uint32_t* sW = (uint32_t*)(smem + region0);          // [4*NC][CPB] packed letters
    const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
    const int tile = blockIdx.x;
    const int b0 = blockIdx.y * CPB;

    // ---- stage the slab (coalesced 16 B per lane) and the chains' window letters
    const float4* src = a.Jt + (size_t)tile * NC * 320;
    for (int k = tid; k < NC * 320; k += 256) sT[k] = src[k];
    const int words = 4 * NC;
    for (int w = tid; w < CPB * words; w += 256) {
        int cl = w / words, wd = w - cl * words;
        int b = b0 + cl;
        uint32_t v = 0;
        if (b < a.n) v = *(const uint32_t*)(a.idx + (size_t)b * g.Ls + g.sh + g.i0 + 4 * wd);
        sW[wd * CPB + cl] = v;
    }
    __syncthreads();

    // ---- gather: wave = part, lane = chain
    float4 acc[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ck = 0; ck < NC; ++ck) {
        const float4* rows = sT + ((ck * 4 + part) * 4) * 20;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            uint32_t w = sW[(part * NC + ck) * CPB + gi * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t letter = min((w >> (8 * q)) & 0xFFu, 19u);
                float4 r = rows[q * 20 + letter];
                acc[gi].x += r.x; acc[gi].y += r.y; acc[gi].z += r.z; acc[gi].w += r.w;
            }
        }
    }
    __syncthreads();                                     // everyone is done with the slab
    float4* sR = smem;                                   // [4][CPB] part sums (aliases the slab)
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) sR[part * CPB + gi * 64 + lane] = acc[gi];
    __syncthreads();

    // ---- combine parts, add fields, write gradient columns and the energy term
    const int i = tile / 5, k0 = (tile % 5) * 4;
    const float4 h4 = *(const float4*)(a.h + i * 20 + k0);
    const int wpart = i / (4 * NC), wrem = i - wpart * 4 * NC;   // where residue i sits in sW
    for (int cl = tid; cl < CPB; cl += 256) {
        int b = b0 + cl;
        if (b >= a.n) continue;
        float4 s0 = sR[cl], s1 = sR[CPB + cl], s2 = sR[2 * CPB + cl], s3 = sR[3 * CPB + cl];
        float4 S;
        S.x = (s0.x + s1.x) + (s2.x + s3.x);
        S.y = (s0.y + s1.y) + (s2.y + s3.y);
        S.z = (s0.z + s1.z) + (s2.z + s3.z);
        S.w = (s0.w + s1.w) + (s2.w + s3.w);
        int slot = a.slot_fixed;
        if (a.slot_mode == 1) slot = (a.cursel[b] == 0) ? 1 : 0;
        float4* out = (float4*)(a.grad + ((size_t)slot * a.n + b) * g.N + (g.i0 + i) * 20 + k0);
        float4 o = make_float4(S.x + h4.x, S.y + h4.y, S.z + h4.z, S.w + h4.w);
        if (a.accumulate) {
            float4 p = *out;
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *out = o;
        uint32_t letter = (sW[(wpart * NC + (wrem >> 2)) * CPB + cl] >> (8 * (wrem & 3))) & 0xFFu;
        int kk = (int)letter - k0;
        if (kk >= 0 && kk < 4) {
            float sv = kk == 0 ? S.x : kk == 1 ? S.y : kk == 2 ? S.z : S.w;
            float hv = kk == 0 ? h4.x : kk == 1 ? h4.y : kk == 2 ? h4.z : h4.w;
            a.epart[((size_t)slot * a.n + b) * g.Lp + i] = hv + 0.5f * sv;
        }
    }
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
