// Supervised expert on gfx950: forward + input-gradient of one OnehotCNN per (chain, network) workgroup.
//
// Replaces OnehotCNN.forward (reference ppde/nets.py:363-376), the ensemble mean's per-network terms
// (ppde/nets.py:434-442) and autograd of them (ppde/energy.py:108):
//   pre1[t,o] = bc[o] + sum_kappa Wc[o, a_{t+kappa}, kappa]      conv1d on a one-hot input = 5 table rows
//   h1 = relu(pre1);  pre2 = h1 We^T + be;  h2 = relu(pre2);  m[f] = max_t h2[t,f] (first arg-max t*)
//   out = bd + wd . m
//   d out/d x[p,c] = sum_kappa sum_o [pre1>0][p-kappa,o] * ( sum_{f: t*_f = p-kappa, m_f>0} wd_f We[f,o] ) * Wc[o,c,kappa]
// h1 and the routed gradient live in LDS; both dense contractions ([T x C] x [C x F] and [T x C] x [C x 5*20])
// run as register-blocked fp32 FMA with the activations broadcast from LDS. Arithmetic per chain does not
// depend on the batch, so sharding chains over GPUs cannot change a bit.
#pragma once
#include "common.h"

struct CnnNet {
    const float* WcT;    // [K][20][CP]   conv weights, table form (channel-contiguous), zero padded
    const float* bc;     // [CP]
    const float4* WeT4;  // [CP/4][F][4]  embedding weights, k-interleaved for 16-byte lane loads
    const float* We;     // [F][CP]
    const float* be;     // [F]
    const float* wd;     // [F]
    const float4* Wf4;   // [CP/4][J][4]  conv weights as [C] x [kappa*20 + c], k-interleaved
    float bd;
};

struct CnnArgs {
    CnnNet net[4];
    int n_nets;
    int C, CP, K, F, T, J;          // J = K*20
    const uint8_t* idx;             // states [n][Ls]
    float* gradC;                   // [slots][nets][n][N]
    float* fitC;                    // [slots][nets][n]
    int slot;                       // evaluation slot to write
    int n;                          // chains in the buffers (slot stride)
    int b_off;                      // first chain of this launch
    int want_grad;
    float scale;                    // upstream gradient of every network output: lamda / nets (or 1 / nets)
    Geom g;
};

#define CNN_TB 16

__host__ __device__ inline size_t cnn_lds_bytes(int T, int CP, int F, int J, int L) {
    size_t rows = (size_t)((T + CNN_TB - 1) / CNN_TB) * CNN_TB;
    size_t a = rows * (CP + 4) * 4;                    // h1 (later: routed output O, needs rows*J <= rows*(CP+4))
    size_t o = rows * (size_t)((J > CP + 4) ? J : (CP + 4)) * 4;
    return o + a + (size_t)F * 8 + 64 + ((L + 15) & ~15);
}

// acc[r] += sum_o A[t0+r][o] * W[o][col]   for r < CNN_TB, with A broadcast from LDS and the weight column
// streamed as k-interleaved float4s (W4[(o/4)][col]).
__device__ __forceinline__ void fma_block(float (&acc)[CNN_TB], const float* A, int AS, int t0,
                                          const float4* W4, int ncols, int col, int CP) {
    for (int o4 = 0; o4 < CP / 4; ++o4) {
        const float4 w = W4[(size_t)o4 * ncols + col];
#pragma unroll
        for (int r = 0; r < CNN_TB; ++r) {
            const float4 h = *(const float4*)(A + (t0 + r) * AS + 4 * o4);
            acc[r] = fmaf(h.x, w.x, acc[r]);
            acc[r] = fmaf(h.y, w.y, acc[r]);
            acc[r] = fmaf(h.z, w.z, acc[r]);
            acc[r] = fmaf(h.w, w.w, acc[r]);
        }
    }
}

__global__ __launch_bounds__(256) void k_cnn(CnnArgs a) {
    extern __shared__ unsigned char smem_raw[];
    const Geom g = a.g;
    const int b = a.b_off + blockIdx.x, ni = blockIdx.y, tid = threadIdx.x;
    const CnnNet net = a.net[ni];
    const int T = a.T, CP = a.CP, F = a.F, J = a.J, K = a.K;
    const int AS = CP + 4;                                     // LDS row stride of h1 / dH1
    const int rows = ((T + CNN_TB - 1) / CNN_TB) * CNN_TB;
    const int OS = (J > AS) ? J : AS;
    float* sH = (float*)smem_raw;                               // [rows][AS]  h1, later O [rows][J]
    float* sD = sH + (size_t)rows * OS;                         // [rows][AS]  routed gradient
    float* sM = sD + (size_t)rows * AS;                         // [F] max values
    int* sTs = (int*)(sM + F);                                  // [F] arg-max rows
    float* red = (float*)(sTs + F);                             // 8 floats
    uint8_t* sSt = (uint8_t*)(red + 16);                        // [L] letters
    int phase = 0;

    const int slot = a.slot;

    for (int l = tid; l < g.L; l += 256) sSt[l] = min((int)a.idx[(size_t)b * g.Ls + g.sh + l], 19);
    __syncthreads();

    // ---- h1 = relu(conv): five table rows per (t, channel); padded rows/channels are zero
    for (int e = tid; e < rows * AS; e += 256) {
        const int t = e / AS, o = e - t * AS;
        float v = 0.f;
        if (t < T && o < CP) {
            v = net.bc[o];
            for (int kp = 0; kp < K; ++kp) v += net.WcT[((size_t)kp * 20 + sSt[t + kp]) * CP + o];
            v = fmaxf(v, 0.f);
        }
        sH[t * AS + o] = v;
        sD[t * AS + o] = 0.f;
    }
    __syncthreads();

    // ---- pre2 = h1 We^T + be, relu, running max over t (strict >: first index wins, like torch.max)
    for (int f = tid; f < F; f += 256) {
        const float bias = net.be[f];
        float m = -INFINITY;
        int ts = 0;
        for (int t0 = 0; t0 < rows; t0 += CNN_TB) {
            float acc[CNN_TB];
#pragma unroll
            for (int r = 0; r < CNN_TB; ++r) acc[r] = bias;
            fma_block(acc, sH, AS, t0, net.WeT4, F, f, CP);
#pragma unroll
            for (int r = 0; r < CNN_TB; ++r) {
                const float v = fmaxf(acc[r], 0.f);
                if (t0 + r < T && v > m) { m = v; ts = t0 + r; }
            }
        }
        sM[f] = m;
        sTs[f] = ts;
    }
    __syncthreads();

    // ---- out = bd + wd . m  (fixed tree)
    {
        float s = 0.f;
        for (int f = tid; f < F; f += 256) s += net.wd[f] * sM[f];
        const float tot = block_sum<4>(s, red, phase);
        if (tid == 0) a.fitC[((size_t)slot * a.n_nets + ni) * a.n + b] = tot + net.bd;
    }
    if (!a.want_grad) return;

    // ---- route: dH1[t*_f][o] += scale * wd_f * We[f][o] for every feature whose max is positive;
    //      thread = channel, features in index order (deterministic)
    for (int o = tid; o < CP; o += 256) {
        for (int f = 0; f < F; ++f) {
            if (sM[f] > 0.f) {
                const float c = a.scale * net.wd[f];
                sD[sTs[f] * AS + o] += c * net.We[(size_t)f * CP + o];
            }
        }
    }
    __syncthreads();
    // ---- gate by the first ReLU
    for (int e = tid; e < rows * AS; e += 256) {
        if (!(sH[e] > 0.f)) sD[e] = 0.f;
    }
    __syncthreads();
    // ---- O[t][kappa*20 + c] = sum_o dpre1[t][o] Wc[o][c][kappa]   (O aliases h1)
    float* sO = sH;
    for (int w = tid; w < J * (rows / CNN_TB); w += 256) {
        const int j = w % J, t0 = (w / J) * CNN_TB;
        float acc[CNN_TB];
#pragma unroll
        for (int r = 0; r < CNN_TB; ++r) acc[r] = 0.f;
        fma_block(acc, sD, AS, t0, net.Wf4, J, j, CP);
#pragma unroll
        for (int r = 0; r < CNN_TB; ++r) sO[(t0 + r) * OS + j] = acc[r];
    }
    __syncthreads();
    // ---- transposed convolution: dx[p][c] = sum_kappa O[p - kappa][kappa*20 + c]
    float* out = a.gradC + (((size_t)slot * a.n_nets + ni) * a.n + b) * g.N;
    for (int e = tid; e < g.N; e += 256) {
        const int p = e / 20, c = e - 20 * p;
        float v = 0.f;
        for (int kp = 0; kp < K; ++kp) {
            const int t = p - kp;
            if (t >= 0 && t < T) v += sO[t * OS + kp * 20 + c];
        }
        out[e] = v;
    }
}
