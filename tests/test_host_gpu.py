"""GPU tests of the host layer around the kernels: hipGraph bookkeeping, scratch ownership, argument checks,
the shared device copy of the couplings."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ppde_amd import _hip, synthetic
from ppde_amd.encoding import seqs_to_idx


def _model(L, Lp, i0, with_cnn, lam, seed=5):
    from ppde_amd.energy import HipModel
    rng = np.random.default_rng(seed)
    wt = rng.integers(0, 20, L).astype(np.uint8)
    J, h = synthetic.make_potts(Lp, seed=seed)
    m = HipModel(wt, "cuda:0")
    m.set_potts(J, h, i0)
    if with_cnn:
        m.set_cnn([synthetic.make_cnn_state(L, s) for s in range(3)])
    m.set_lamda(lam)
    return m, wt


def _run(m, wt, n, T, blocks, which, use_graph, Lp, i0, reuse=True, seed=9):
    from ppde_amd.sampler import Chains
    ch = Chains(m, n, T, 2, 3, False, i0, i0 + Lp - 1, which, 1, reuse_grad=reuse, random_chain=0, use_graph=use_graph, seed=seed)
    ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
    for k in blocks:
        ch.run(k)
    return ch, ch.collect()


def test_graphs_are_captured_by_init_and_replayed_by_every_run():
    """The driver's shape (--warmup 5 --steps 20): the warm-up is shorter than a segment and runs eagerly, the timed
    block is one replay; nothing is captured inside a run. Replay and eager launches give the same bits."""
    m, wt = _model(96, 80, 8, False, 0.0)
    ch, res = _run(m, wt, 16, 160, [5, 20, 20, 100, 7], 1, True, 80, 8)
    st = ch.graph_stats()
    assert st["captures"] == 2 and st["captures_in_run"] == 0        # segments of 100 and 20 iterations
    assert st["replayed_steps"] == 140 and st["eager_steps"] == 12
    _, ref = _run(m, wt, 16, 160, [152], 1, False, 80, 8)
    assert np.array_equal(res["energy_history"], ref["energy_history"]) and np.array_equal(res["best_idx"], ref["best_idx"])
    ch2, _ = _run(m, wt, 4, 30, [30], 1, True, 80, 8)                  # histories too short for the 100-step segment
    st2 = ch2.graph_stats()
    assert st2["captures"] == 1 and st2["replayed_steps"] == 20 and st2["eager_steps"] == 10


def test_chunk_scratch_belongs_to_the_chains_object():
    """L = 104 takes the chunked CNN kernels. A second, larger population (or a large stateless call) on the SAME model
    must not disturb the first one's captured graphs (its chunk scratch used to live in the model and was
    reallocated)."""
    m, wt = _model(104, 76, 23, True, 2.0)
    _, ref = _run(m, wt, 6, 60, [60], 3, True, 76, 23)
    from ppde_amd.sampler import Chains
    a = Chains(m, 6, 60, 2, 3, False, 23, 98, 3, 1, reuse_grad=True, random_chain=0, use_graph=True, seed=9)
    a.init(torch.as_tensor(np.tile(wt, (6, 1))).cuda())
    a.run(20)
    b = Chains(m, 40, 60, 2, 3, False, 23, 98, 3, 1, reuse_grad=True, random_chain=0, use_graph=True, seed=10)
    b.init(torch.as_tensor(np.tile(wt, (40, 1))).cuda())
    b.run(20)
    m.energy_grad(torch.as_tensor(np.tile(wt, (100, 1))).cuda(), 3)
    a.run(40)
    b.run(40)
    res = a.collect()
    assert np.array_equal(res["energy_history"], ref["energy_history"]) and np.array_equal(res["best_idx"], ref["best_idx"])
    assert np.isfinite(b.collect()["energy_history"]).all()


def test_path_length_beyond_the_supplied_noise_is_an_error():
    from ppde_amd.sampler import Chains
    m, wt = _model(24, 16, 4, False, 0.0)
    n, N = 4, 24 * 20
    ch = Chains(m, n, 4, 2, 0, False, 4, 19, 1, 0)
    ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
    U = torch.tensor([[1, 3, 1, 1]], dtype=torch.int32)              # chain 1 wants 3 sub-steps ...
    q = torch.ones(1, n, N)                                          # ... but one sub-step of variates is supplied
    with pytest.raises(_hip.PpdeHipError, match="max_u"):
        ch.run(1, (U, q, torch.zeros(1, n), [1]))


def test_one_device_copy_of_the_couplings():
    """Energy function, ground-truth model and Potts score share one ppde_model when they read the same potts.pkl."""
    import argparse
    from ppde_amd.energy import ProteinProductOfExperts
    from ppde_amd.nets import AugmentedLinearRegression, proteins_potts_score
    with tempfile.TemporaryDirectory() as root:
        synthetic.write_weights_dir(root, "TOY24", potts_seed=7)
        args = argparse.Namespace(energy_lamda=5.0, unsupervised_expert="potts", protein_weights=root, protein="TOY24",
                                  n_chains=4, device="cuda:0", ppde_rng="philox")
        en = ProteinProductOfExperts(args)
        alr = AugmentedLinearRegression(os.path.join(root, "TOY24"), "cuda:0")
        assert alr.model is en.model
        x = en.wt_onehot.repeat(3, 1, 1)
        s = proteins_potts_score(x, os.path.join(root, "TOY24"))
        assert float(s.abs().max()) == 0.0
        assert torch.isfinite(alr(x)).all()


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from ppde_amd import parallel, synthetic
from ppde_amd.encoding import seqs_to_idx
assert parallel.active() and dist.get_backend() == "nccl"
# host tensor -> device staging -> all_gather over RCCL -> back to the host, along both axes, three dtypes
for t, dim in ((torch.arange(12, dtype=torch.uint8).reshape(3, 4), 0), (torch.randn(5, 3), 1), (torch.arange(7, dtype=torch.int32), 0)):
    out = parallel.all_gather_rows(t, t.shape[dim], dim=dim)
    assert out.device.type == "cpu" and out.dtype == t.dtype and torch.equal(out, t), (t.dtype, dim)
assert torch.equal(parallel.broadcast_from(torch.full((3,), 4.0), 0), torch.full((3,), 4.0))
assert parallel.agree_from_rank0([123, 2 ** 62 + 5]) == [123, 2 ** 62 + 5]
ones = torch.ones(1, device="cuda:0")
dist.all_reduce(ones)
dist.barrier()
assert float(ones) == 1.0
# the sampler's sharded collect (PPDE_PAS.run(..., ppde_shard=True)) through the same path == the plain run
import argparse
from ppde_amd.energy import HipModel
from ppde_amd.sampler import PPDE_PAS
_, seq, (i0, Lp) = synthetic.PROTEINS["TOY24"]
wt = seqs_to_idx([seq])[0]
J, h = synthetic.make_potts(Lp, seed=7)
class E:                      # minimal energy-function shell around a HipModel
    which = 1
E.model = HipModel(wt, "cuda:0"); E.model.set_potts(J, h, i0)
x0 = torch.nn.functional.one_hot(torch.as_tensor(np.tile(wt, (9, 1))).long(), 20).float()
res = []
for shard in (False, True):
    a = argparse.Namespace(ppde_pas_length=2, nmut_threshold=3, paper_results=False, ppde_rng="philox", ppde_seed=11, ppde_shard=shard)
    np.random.seed(1)
    res.append(PPDE_PAS(a).run(x0, 30, E, i0, i0 + Lp - 1, lambda x: torch.zeros(x.shape[0]), log_every=10))
assert torch.equal(res[0][0], res[1][0]) and all(np.array_equal(p, q) for p, q in zip(res[0][1:5], res[1][1:5]))
assert all(np.array_equal(p, q) for p, q in zip(res[0][5], res[1][5]))
dist.destroy_process_group()
print("rccl one-rank ok")
"""


def test_rccl_path_executes_on_one_gpu():
    """A process group of ONE rank on the nccl (= RCCL) backend with the world-size-1 shortcut switched off: RCCL loads,
    builds a communicator and moves the population collect's buffers (host -> device staging -> all_gather / broadcast ->
    host), and `bench.py --gpus 1` runs its nccl branch (init, barrier, all_reduce, timed gather). Every other multi-rank
    test uses gloo; this is the only place the RCCL code path runs before an 8-GPU node sees it."""
    import json
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PPDE_COLLECTIVES_AT_WORLD_1="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(_RCCL_ONE_RANK)
        path = fh.name
    try:
        r = subprocess.run([sys.executable, path, REPO], capture_output=True, text=True, timeout=420, env=env)
        assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    finally:
        os.unlink(path)
    env.update(MASTER_PORT="29548", PPDE_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "2",
                        "--no-cpu-baseline", "--no-large"], capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["rccl_ranks"] == 1 and line["population_gather_ms"] > 0 and line["value"] > 0
