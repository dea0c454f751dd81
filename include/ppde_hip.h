/*
 * ppde_hip.h — C ABI of the MI355X (gfx950) PPDE sampler hot path.
 *
 * The reference (pemami4911/ppde) has no FFI: its boundary for this path is the duck-typed Python API
 *   ppde/base_sampler.py:8-15            BaseSampler.run(...)
 *   ppde/protein_samplers/ppde.py:24-192 PPDE_PAS.run(...)
 *   ppde/energy.py:97-140                ProteinProductOfExperts.get_energy / get_energy_and_grads / ...
 * This header is what a ctypes/cffi binding of that API calls (ppde_amd/_hip.py is that binding; the stub a
 * maintainer of the reference would add is shown in INTEGRATION.md). Each entry point names the reference
 * code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ppde_status otherwise; ppde_last_error() returns a
 *     thread-local human-readable message for the last failure; nothing throws across the ABI;
 *   - "host" pointers are ordinary host memory owned by the caller; "dev" pointers are device memory on the
 *     model's device (e.g. torch tensors' data_ptr()); the library owns every device buffer it allocates;
 *   - states are residue indices, one byte per residue (alphabet ACDEFGHIKLMNPQRSTVWY -> 0..19,
 *     ppde/third_party/hsu/data_utils.py:48-70); fp32 one-hot [n, L, 20] exists only at the API edge;
 *   - a ppde_model is immutable after its set_* calls and may be shared by several ppde_chains; a
 *     ppde_chains owns one HIP stream (and every scratch buffer its kernels and captured graphs touch)
 *     and is not thread-safe; the stateless calls (ppde_energy_grad ...) share one scratch set per model
 *     and must not run concurrently with each other on the same model.
 *
 * Shape limits (checked; violations return PPDE_ERR_INVALID with a message, nothing falls back):
 *   - alphabet 20; sequence length 5 <= L <= 4096 for the model, L <= 307 for ppde_chains (a chain's
 *     L*20 proposal logits live in the registers of one 512-thread workgroup);
 *   - state rows (L + alignment padding) <= 512 bytes for the Potts kernel, i.e. L <= ~500;
 *   - Potts window 1 <= Lp <= L; windows of more than 128 residues stream through an LDS ring
 *     (no upper bound besides the state-row limit), shorter ones keep a resident slab of <= 32 KiB
 *     pieces per wave;
 *   - supervised expert: 1..4 networks of one shape, kernel size 1..8, embedding width F <= 512
 *     for the single-launch kernel (wider or longer networks take the chunked kernels);
 *   - transformer expert: head width 24, 32 or 64, dim a multiple of 8 (<= 1536; padded to a multiple of 128 internally), ffn a
 *     multiple of 128, L <= 256;
 *   - ppde_pas_length 1..64; chain_offset + n_chains < 2^32.
 */
#ifndef PPDE_HIP_H
#define PPDE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPDE_ABI_VERSION 1
#define PPDE_ALPHABET 20
/* `which` bit 3, see ppde_energy_grad */
#define PPDE_WHICH_FULL_GRAD 8

typedef enum {
    PPDE_OK = 0,
    PPDE_ERR_INVALID = -1,     /* bad argument / shape / state */
    PPDE_ERR_HIP = -2,         /* a HIP runtime call failed */
    PPDE_ERR_NOT_ONEHOT = -3,  /* an input row is not a one-hot vector */
    PPDE_ERR_NUMERIC = -4      /* a proposal row had no finite logit (the reference raises ValueError there) */
} ppde_status;

typedef struct ppde_model ppde_model;
typedef struct ppde_chains ppde_chains;

int ppde_abi_version(void);
const char* ppde_last_error(void);
/* number of visible HIP devices, or a negative status */
int ppde_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Model = the product of experts' parameters resident on one device.
 * Replaces the constructors ProteinProductOfExperts.__init__ (ppde/energy.py:72-95),
 * PottsModel.__init__ (ppde/nets.py:245-262) and EnsembleProtein.__init__ (ppde/nets.py:417-424).
 * ---------------------------------------------------------------------------------------------- */

/* L = full sequence length; wt_idx host [L] = wild-type residues (energy.py:95 `wt_onehot`). */
int ppde_model_create(ppde_model** out, int device, int L, const uint8_t* wt_idx);
int ppde_model_destroy(ppde_model* m);

/* Potts couplings J host [Lp, Lp, 20, 20] and fields h host [Lp, 20] (potts.pkl `J_ij`, `h_i`,
 * nets.py:247-249); the window covers residues [win_start, win_start + Lp) (nets.py:280).
 * J need not be symmetric: it is symmetrised on upload, M = (J + J^T)/2, which is what autograd of
 * nets.py:287-290 yields (energy.py:108). Also evaluates wt_H (nets.py:262). */
int ppde_model_set_potts(ppde_model* m, const float* J, const float* h, int Lp, int win_start);

/* Supervised CNN ensemble (nets.py:350-376, :412-442): n_nets networks, each
 * conv_w [C, 20, K], conv_b [C], lin_w [F, C], lin_b [F], dec_w [F], dec_b [1] on the host. */
int ppde_model_set_cnn(ppde_model* m, int n_nets, int C, int K, int F,
                       const float* const* conv_w, const float* const* conv_b,
                       const float* const* lin_w, const float* const* lin_b,
                       const float* const* dec_w, const float* const* dec_b);

/* Transformer unsupervised expert (nets.py:172-240 `Transformer`, :302-312 `PottsTransformer`): an ESM-2 encoder
 * evaluated on one-hot input, score = sum x * log_softmax(logits) minus the wild type's (nets.py:219-240), under the
 * reference's autocast (fp16 matmuls, fp32 statistics). The reference takes the model from the third-party
 * `esm_one_hot` package + torch hub; here the caller passes ESM-2's parameters (fp32, host), named as in
 * facebookresearch/esm's ESM2: per-layer arrays have n_layers entries. Written for head widths 24, 32 and 64
 * (esm2_t12_35M: dim 480, 20 heads, ffn 1920; esm2_t30_150M: 640, 20, 2560; esm2_t33_650M: 1280, 20, 5120), ffn a multiple
 * of 128, L <= 256. Also evaluates the wild
 * type's score. */
typedef struct {
    const float* embed;                 /* embed_tokens.weight [33][dim] (also the tied LM-head projection) */
    const float* const* q_w; const float* const* q_b;       /* layers.i.self_attn.{q,k,v,out}_proj.{weight [dim][dim], bias} */
    const float* const* k_w; const float* const* k_b;
    const float* const* v_w; const float* const* v_b;
    const float* const* o_w; const float* const* o_b;
    const float* const* ln1_w; const float* const* ln1_b;   /* layers.i.self_attn_layer_norm */
    const float* const* ln2_w; const float* const* ln2_b;   /* layers.i.final_layer_norm */
    const float* const* fc1_w; const float* const* fc1_b;   /* [ffn][dim], [ffn] */
    const float* const* fc2_w; const float* const* fc2_b;   /* [dim][ffn], [dim] */
    const float* final_ln_w; const float* final_ln_b;       /* emb_layer_norm_after */
    const float* head_dense_w; const float* head_dense_b;   /* lm_head.dense */
    const float* head_ln_w; const float* head_ln_b;         /* lm_head.layer_norm */
    const float* head_bias;                                 /* lm_head.bias [33] */
} ppde_tf_weights;
int ppde_model_set_transformer(ppde_model* m, int n_layers, int dim, int heads, int ffn, const ppde_tf_weights* w);
/* local score of the wild type (nets.py:188 `wt_score`) for inspection. */
int ppde_model_get_transformer_wt_score(ppde_model* m, float* out_host);

/* Timing hook for bench.py: average duration (microseconds) of the transformer's GEMM kernel, C[M,N] = A[M,K] B[N,K]^T,
 * fp16 operands filled with pseudo-random values, `reps` launches between one HIP event pair. epilogue: 0 bias,
 * 2 bias + residual, 3 bias + GELU (the fc1 form), 4 GELU' (backward through fc2), 5 plain. M, N multiples of 128,
 * K of 64. */
int ppde_transformer_time_gemm(int device, int M, int N, int K, int reps, int epilogue, float* avg_us);

/* Diagnostics for the parity tests: an fp16 activation of the last stateless evaluation that used the transformer
 * expert, converted to fp32. what: 0 layer input, 1 q|k|v, 2 (not kept: the backward rebuilds the attention probabilities), 3 post-attention stream,
 * 4 GELU' of the fc1 pre-activation (what the forward keeps for the backward; all of `layer`), 5 final stream, 6 logits, 7 d logits, 8 d embedding, 9 d tokens,
 * 10 / 11 attention output / d q|k|v of the layer evaluated last. The workspace holds ONE chunk of chains (PPDE_TF_WORK_GB; after an
 * evaluation in several chunks: the last one): `count` beyond the buffer's padded rows x width is PPDE_ERR_INVALID. */
int ppde_debug_transformer_read(ppde_model* m, int what, int layer, float* out_host, int64_t count);

/* The fc1 GEMM timed IN SITU for bench.py's roofline object: one stateless transformer evaluation (energy + gradient)
 * of idx_dev [n, L] with a HIP event pair around every fc1 launch (real activations, real neighbouring kernels); mean
 * event-to-event time in microseconds and the number of launches timed (= layers). */
int ppde_transformer_time_fc1_in_situ(ppde_model* m, const uint8_t* idx_dev, int n, float* avg_us, int* launches);

/* lamda of e = dH + lamda * fit (energy.py:74, :99-100). */
int ppde_model_set_lamda(ppde_model* m, float lamda);

/* wt_H (nets.py:262) for inspection. */
int ppde_model_get_wt_hamiltonian(ppde_model* m, float* out_host);

/* ------------------------------------------------------------------------------------------------
 * Stateless evaluations at the API edge (device pointers; `stream` is a hipStream_t or NULL).
 * ---------------------------------------------------------------------------------------------- */

/* fp32 one-hot x_dev [n, L, 20] -> idx_dev [n, L]. Fails with PPDE_ERR_NOT_ONEHOT (after a stream sync)
 * if some row is not exactly one 1.0 and nineteen 0.0. */
int ppde_onehot_to_idx(ppde_model* m, const float* x_dev, int n, uint8_t* idx_dev, void* stream);
/* idx_dev [n, L] -> fp32 one-hot x_dev [n, L, 20]. */
int ppde_idx_to_onehot(ppde_model* m, const uint8_t* idx_dev, int n, float* x_dev, void* stream);

/* get_energy / get_energy_and_grads (energy.py:97-132): e_dev [n], fit_dev [n], and, when grad_dev
 * is not NULL, grad_dev [n, L, 20] = d e.sum() / d x. which: bit 0 = Potts expert, bit 1 = supervised expert,
 * bit 2 = transformer expert; 3 = Potts product of experts, 6 = transformer product of experts
 * (`--unsupervised_expert transformer`), 7 = `potts+transformer`. With which == 2, e = fit and grad = d fit/dx
 * (ProteinSupervised, energy.py:153-160); without bit 1, fit = 0.
 * The gradient follows the reference branch by branch: for the Potts product of experts (3) it is
 * d(dH + lamda * fit)/dx (energy.py:105-108); with the transformer expert (6, 7) the reference computes fit from x
 * but differentiates with respect to the minibatch SLICE of x (energy.py:115, :125), so lamda * d fit/dx never
 * reaches grad_x: grad = d(unsupervised experts)/dx only, while e still holds + lamda * fit (and no CNN backward
 * runs). Bit 3 (PPDE_WHICH_FULL_GRAD, `args.ppde_full_grad`) opts into d e/dx with the supervised term for 6 / 7. */
int ppde_energy_grad(ppde_model* m, const uint8_t* idx_dev, int n, int which,
                     float* e_dev, float* fit_dev, float* grad_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Chains = the sampler state of n independent Markov chains (PPDE_PAS.run, ppde.py:24-192).
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    int32_t n_chains;        /* chains held by this object (this rank's shard) */
    int32_t max_steps;       /* capacity of the histories (T); run may stop earlier */
    int32_t pas_length;      /* args.ppde_pas_length (ppde.py:12): U ~ randint[1, 2*pas) */
    int32_t nmut_threshold;  /* args.nmut_threshold; 0 = unlimited (ppde.py:15-17) */
    int32_t paper_results;   /* args.paper_results (ppde.py:14,76,148) */
    int32_t min_pos;         /* proposals restricted to residues [min_pos, max_pos] (ppde.py:60-63) */
    int32_t max_pos;
    int32_t which;           /* experts in the energy (+ PPDE_WHICH_FULL_GRAD), as in ppde_energy_grad */
    int32_t rng_mode;        /* 0 = noise supplied by the caller per step (parity with a host generator),
                                1 = counter-based Philox4x32-10 on the device */
    int32_t reuse_grad;      /* 1 = keep energy/gradient of the current state from the previous iteration
                                instead of re-evaluating it (identical results, half the expert calls) */
    int32_t record_after_reset; /* 1 = histories of STATES hold the post-reset state (the reference's
                                --device cpu aliasing artefact); 0 = pre-reset (its cuda behaviour) */
    int32_t trace;           /* 1 = keep per-iteration draws / accept bits / log-acceptance for tests */
    int32_t random_chain;    /* local index of the chain whose trajectory is kept (ppde.py:37,47,142), or -1 */
    int32_t use_graph;       /* 1 = replay iterations from hipGraphs captured at ppde_chains_init (rng_mode 1 only) */
    int32_t n_streams;       /* >1: the chains are cut into this many sub-populations whose iterations run on
                                separate HIP streams and overlap on the GPU (chains are independent, results
                                are unchanged); 0/1 = one stream. rng_mode 1 only. */
    uint64_t seed;           /* Philox key */
    uint64_t chain_offset;   /* global index of local chain 0: results do not depend on the sharding */
} ppde_chain_config;

int ppde_chains_create(ppde_chains** out, ppde_model* m, const ppde_chain_config* cfg);
int ppde_chains_destroy(ppde_chains* c);

/* Start from idx0_dev [n, L] (ppde.py:35-47): evaluates the initial energies, fills history row 0. */
int ppde_chains_init(ppde_chains* c, const uint8_t* idx0_dev);

/* Run `steps` iterations of ppde.py:65-153. For rng_mode 0 the caller supplies, for these iterations,
 *   U_dev int32 [steps, n]          path lengths in [1, 2*pas)        (torch.randint, ppde.py:67)
 *   q_dev fp32  [sum_t max_u(t), n, L*20]  Exp(1) variates, max_u(t) = max_b U[t,b]  (multinomial, ppde.py:109)
 *   u_dev fp32  [steps, n]          accept uniforms                  (torch.rand_like, ppde.py:138)
 *   max_u host int32 [steps]
 * and for rng_mode 1 passes NULLs. Asynchronous: returns once the work is enqueued. */
int ppde_chains_run(ppde_chains* c, int steps, const int32_t* U_dev, const float* q_dev,
                    const float* u_dev, const int32_t* max_u);

/* Block until enqueued work is done; reports PPDE_ERR_NUMERIC if a proposal row degenerated. */
int ppde_chains_sync(ppde_chains* c);

int ppde_chains_steps_done(ppde_chains* c);

/* Current population (after the mutation-cap reset) and what the reference logs every log_every
 * (ppde.py:155-170): any pointer may be NULL. idx host [n, L], energy/fitness host [n] = last history
 * row, accepted host [n] = last accept bits, dist host [n] = mutation counts. Synchronises. */
int ppde_chains_peek(ppde_chains* c, uint8_t* idx, float* energy, float* fitness, uint8_t* accepted, int32_t* dist);

/* Final collect (ppde.py:172-192): best state per chain by max energy over history (first index on ties),
 * histories [steps_done+1, n], random trajectory [steps_done+1, L]. Any pointer may be NULL. Synchronises. */
int ppde_chains_collect(ppde_chains* c, uint8_t* best_idx, float* best_energy, float* best_fitness,
                        int32_t* best_step, float* energy_history, float* fitness_history, uint8_t* random_traj);

/* hipGraph bookkeeping for bench.py: graphs are captured and instantiated by ppde_chains_init (segments of
 * 100 and 20 iterations when max_steps allows, or PPDE_GRAPH_LEN), never inside ppde_chains_run;
 * captures_in_run counts violations of that (always 0), replayed/eager_steps say how the iterations of all runs
 * so far were issued. Any pointer may be NULL. */
int ppde_chains_graph_stats(ppde_chains* c, int32_t* captures, int32_t* captures_in_run,
                            int64_t* replayed_steps, int64_t* eager_steps);

/* Trace buffers (cfg.trace = 1): flat host int32 [steps_done, 2*pas-1, n] (-1 = not drawn),
 * accepted host uint8 [steps_done, n], log_acc host fp32 [steps_done, n], U host int32 [steps_done, n]. */
int ppde_chains_trace(ppde_chains* c, int32_t* flat, uint8_t* accepted, float* log_acc, int32_t* U);

/* Device RNG inspection (tests): fills q_dev [n, L*20], u_dev [n], U_dev [n] with what iteration `it`,
 * sub-step `s` would use in rng_mode 1. The device RNG draws the categorical of a sub-step in two levels (residue, then
 * letter: the same law as the reference's flat race, ppde.py:106-110, with L + 20 variates instead of L*20): row b of q_dev
 * holds the Exp(1) variates of the residue race in [0, L), those of the letter race in [L, L + 20), and 1.0 beyond. */
int ppde_chains_philox_dump(ppde_chains* c, int it, int s, float* q_dev, float* u_dev, int32_t* U_dev);

/* Timing hooks for bench.py: average duration in microseconds of the Potts energy+gradient kernel over
 * the launches recorded since the last reset, measured with HIP events on the chains' stream. */
int ppde_chains_time_potts_kernel(ppde_chains* c, int reps, float* avg_us);

/* The same for ALL experts of the chains' energy (cfg.which, with gradients), e.g. the fused Potts + CNN launch of the
 * Potts product of experts: `reps` evaluations of the current states into the proposal slot between one HIP event
 * pair on the chains' stream; average microseconds per evaluation. */
int ppde_chains_time_experts(ppde_chains* c, int reps, float* avg_us);

/* The same kernel timed IN SITU: runs `iters` real iterations of a Potts-only energy (which = 1; eagerly, on the chains'
 * stream) with a stop event bound to every kernel's dispatch; avg_us = mean time from the predecessor kernel's end to the
 * Potts launch's end in microseconds, launches = how many were timed, avg_dispatch_us (may be NULL) = mean of the Potts
 * dispatches' own start -> end stamps (0 if the runtime does not report them). The iterations count towards steps_done.
 * rng_mode 1 only. */
int ppde_chains_time_potts_in_situ(ppde_chains* c, int iters, float* avg_us, int* launches, float* avg_dispatch_us);

#ifdef __cplusplus
}
#endif
#endif /* PPDE_HIP_H */
