"""Host-side noise for rng_mode 0: the SAME random numbers the reference consumes, in the same order.

Per iteration the reference draws (ppde/protein_samplers/ppde.py): torch.randint(1, 2*pas, (n,1)) (:67, always
the CPU generator), then max_u = max(U) times torch.multinomial(p, 1) (:109) — which on the CPU path is
argmax(p / q) with q = empty_like(p).exponential_(1) — then torch.rand_like(log_acc) (:138). Drawing
randint -> max_u x exponential_ -> rand from the same generator state therefore reproduces its stream.
"""
import torch


def draw_iteration(n, N, pas_length, generator=None):
    """-> (U int64 [n], q fp32 [max_u, n, N], u fp32 [n]) on the CPU."""
    U = torch.randint(1, 2 * pas_length, size=(n, 1), generator=generator).reshape(n)
    max_u = int(U.max())
    q = torch.empty(max_u, n, N)
    for s in range(max_u):
        q[s].exponential_(generator=generator)
    u = torch.rand(n, generator=generator)
    return U, q, u


def draw_chunk(k, n, N, pas_length, generator=None, rows=None):
    """Noise of k consecutive iterations, optionally restricted to chain rows [rows[0], rows[1]) AFTER drawing
    the full population's numbers (every rank of a sharded run draws the same global stream).
    -> (U int32 [k, m], q fp32 [sum max_u, m, N], u fp32 [k, m], max_u list[int])"""
    Us, qs, us, mus = [], [], [], []
    lo, hi = rows if rows is not None else (0, n)
    for _ in range(k):
        U, q, u = draw_iteration(n, N, pas_length, generator)
        mus.append(int(q.shape[0]))
        Us.append(U[lo:hi].to(torch.int32))
        qs.append(q[:, lo:hi])
        us.append(u[lo:hi])
    return torch.stack(Us, 0).contiguous(), torch.cat(qs, 0).contiguous(), torch.stack(us, 0).contiguous(), mus
