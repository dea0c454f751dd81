"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares (no compute calls),
host logic (encoding, loaders, noise order, sharding) and the world_size-2 gather over gloo."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import ppde_oracle as orc
from ppde_amd import encoding, noise, parallel, synthetic, weights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ppde_amd import _hip, build
    build.build()
    hdr = open(os.path.join(REPO, "include", "ppde_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(ppde_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ppde_hip.h but not exported"
    assert declared == set(_hip.SIGNATURES), "ppde_amd/_hip.py must bind exactly the header's functions"
    assert _hip.load().ppde_abi_version() == 1


def test_no_kernel_uses_scratch_memory(tmp_path):
    """A struct the compiler could not keep in registers (run-time indexing, a select between members) shows up as
    private-segment memory and costs microseconds per launch: every kernel of the library must use none."""
    import shutil
    from ppde_amd import _hip, build
    build.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("ROCm llvm tools not installed")
    so = shutil.copy(_hip.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], check=True, capture_output=True)
    obj = [p for p in os.listdir(tmp_path) if "gfx950" in p]
    assert len(obj) == 1, os.listdir(tmp_path)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / obj[0])], check=True,
                           capture_output=True, text=True).stdout
    names = re.findall(r"\.name:\s+(\S+)", notes)
    scratch = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
    kernels = [n for n in names if n.startswith("_Z")]
    assert len(kernels) == len(scratch) == len(spills) and len(kernels) >= 40
    bad = [(k, s, v) for k, s, v in zip(kernels, scratch, spills) if s or v]
    assert not bad, bad


def test_product_path_has_no_cpu_fallback():
    from ppde_amd.energy import HipModel
    with pytest.raises(RuntimeError):
        HipModel(np.zeros(24, np.uint8), "cpu")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            HipModel(np.zeros(24, np.uint8), "cuda")
    # nothing in the package or among the scripts imports an oracle (only tests/, __graft_entry__.smoke() and the
    # cpu_baseline legs of bench.py / bench_transformer.py may)
    for top in ("ppde_amd", "scripts"):
        for root, _, files in os.walk(os.path.join(REPO, top)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(root, f)).read()
                    assert "ppde_oracle" not in src and "esm_oracle" not in src, f


def test_encoding_roundtrip_and_alphabet():
    assert encoding.ALPHABET == "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["ACDY", "WWAA"]
    oh = encoding.seqs_to_onehot(seqs)
    assert oh.shape == (2, 4, 20) and oh.dtype.kind == "i" and (oh.sum(-1) == 1).all()
    assert encoding.onehot2seq(oh) == seqs
    assert encoding.seqs_to_idx([]).shape == (0, 0)
    with pytest.raises(KeyError):
        encoding.seqs_to_idx(["AXA"])
    ragged = encoding.seqs_to_idx(["ACD", "A"])            # short rows are padded with index 0 like the reference
    assert ragged.tolist() == [[0, 1, 2], [0, 0, 0]]


def test_weight_files_roundtrip():
    with tempfile.TemporaryDirectory() as root:
        d = synthetic.write_weights_dir(root, "TOY24", potts_seed=7)
        p = weights.PottsParams(d)
        assert p.win_start == 4 and p.seq_len == 16 and p.offset == 11
        J, h = synthetic.make_potts(16, seed=7)
        assert np.array_equal(p.J, J) and np.array_equal(p.h, h)
        st = weights.load_cnn_states(d)
        assert st[1]["embedding.0.weight"].shape == (48, 24)
        assert np.array_equal(st[2]["encoder.weight"], synthetic.make_cnn_state(24, 2)["encoder.weight"])
        lin = weights.load_linear(d)
        assert len(lin) == 20 and lin[0][0].shape == (1 + 24 * 20,)
        seqs, idx = weights.load_wt(d)
        assert seqs[0] == synthetic.PROTEINS["TOY24"][1] and idx.shape == (1, 24)


def test_noise_order_matches_oracle_draws():
    torch.manual_seed(5)
    a = [orc.draw_noise_torch(6, 40, 3) for _ in range(4)]
    torch.manual_seed(5)
    U, q, u, mus = noise.draw_chunk(4, 6, 40, 3)
    k = 0
    for t in range(4):
        assert torch.equal(a[t][0].to(torch.int32), U[t]) and torch.equal(a[t][2], u[t])
        assert mus[t] == a[t][1].shape[0] and torch.equal(a[t][1], q[k:k + mus[t]])
        k += mus[t]
    # a shard draws the global stream and keeps its rows
    torch.manual_seed(5)
    U2, q2, u2, _ = noise.draw_chunk(4, 6, 40, 3, rows=(2, 5))
    assert torch.equal(U2, U[:, 2:5]) and torch.equal(q2, q[:, 2:5]) and torch.equal(u2, u[:, 2:5])


def test_shard_range_covers_everything():
    for n in (1, 7, 128, 1000):
        for ws in (1, 2, 3, 8):
            r = [parallel.shard_range(n, k, ws) for k in range(ws)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(ws - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


_GLOO = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import ppde_oracle as orc
from ppde_amd import parallel, synthetic, noise
from ppde_amd.encoding import seqs_to_idx
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
_, seq, (i0, Lp) = synthetic.PROTEINS["TOY24"]
wt = seqs_to_idx([seq])[0]; L = len(seq); n, T, pas = 5, 6, 2
J, h = synthetic.make_potts(Lp, seed=7)
en = orc.EnergyOracle(orc.PottsOracle(J, h, i0, torch.as_tensor(wt.astype(np.int64))), None, 0.0)
lo, hi = parallel.shard_range(n, rank, ws)
torch.manual_seed(11)
Ug, qg, ug, mus = noise.draw_chunk(T, n, L * 20, pas)             # the global stream ...
torch.manual_seed(11)
U, q, u, mus2 = noise.draw_chunk(T, n, L * 20, pas, rows=(lo, hi))  # ... and this rank's rows of it
off = np.concatenate([[0], np.cumsum(mus)])
loc = orc.run(en, np.tile(wt.astype(np.int64), (hi - lo, 1)), wt,
              lambda t: (U[t].long(), q[off[t]:off[t + 1]], u[t]), T, i0, i0 + Lp - 1, pas, 2, False)
eh = parallel.all_gather_rows(loc["energy_history"], n, dim=1)
bi = parallel.all_gather_rows(loc["best_idx"], n, dim=0)
full = orc.run(en, np.tile(wt.astype(np.int64), (n, 1)), wt,
               lambda t: (Ug[t].long(), qg[off[t]:off[t + 1]], ug[t]), T, i0, i0 + Lp - 1, pas, 2, False)
# (the torch-CPU oracle's matmuls are not bitwise batch-size independent; the HIP kernels are, see the gpu tests)
assert torch.allclose(eh, full["energy_history"], atol=1e-5, rtol=0), "sharded histories differ from the unsharded run"
assert torch.equal(bi, full["best_idx"])
rt = parallel.broadcast_from(torch.full((3,), float(rank)), 1)
assert float(rt[0]) == 1.0
assert parallel.agree_from_rank0([100 + rank, 7 * (rank + 1)]) == [100, 7]   # every rank ends up with rank 0's integers
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_run_gathers_to_the_unsharded_result_gloo():
    """world_size 2 over gloo: each rank runs its block of chains (noise = its rows of the global stream) and the
    final all_gather reproduces the single-process result exactly."""
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(_GLOO)
        path = fh.name
    try:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29541", path, REPO],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("ok") == 2
    finally:
        os.unlink(path)


def test_esm2_checkpoint_in_the_published_layout_loads(tmp_path):
    """The published ESM-2 files pickle an argparse.Namespace under cfg.model next to the tensors (facebookresearch/esm reads
    cfg.model.encoder_layers / encoder_attention_heads from it); torch >= 2.6 refuses that class under its default
    weights-only unpickler. The loader allows exactly that class, strips the 'encoder.sentence_encoder.' / 'encoder.'
    prefixes, drops the tied lm_head.weight, and returns the head count of the file."""
    import argparse
    path = str(tmp_path / "checkpoints" / "esm2_t30_150M_UR50D.pt")
    st = synthetic.write_esm2_checkpoint(path, 2, 128, 4, 256, seed=2)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert isinstance(ck["cfg"]["model"], argparse.Namespace) and ck["cfg"]["model"].encoder_attention_heads == 4
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)          # what the loader used to do
    sd, heads = weights.load_esm2_state(path, with_heads=True)
    assert heads == 4 and set(sd) == set(st) and "lm_head.weight" not in sd
    assert all(np.array_equal(sd[k], st[k]) for k in st)
    # an already stripped dict without cfg still loads (heads unknown)
    torch.save({k: torch.from_numpy(v) for k, v in st.items()}, str(tmp_path / "plain.pt"))
    sd2, heads2 = weights.load_esm2_state(str(tmp_path / "plain.pt"), with_heads=True)
    assert heads2 is None and set(sd2) == set(st)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent starts torch.distributed.run with N ranks and the same
    arguments, and never touches the GPU itself; as a rank it refuses a world size that differs from --gpus."""
    import importlib
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **k):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # BASELINE configs[3] as the reference would run it: GFP, Potts + supervised CNN, 8 ranks; --lamda defaults to the README's 15
    monkeypatch.setattr(sys, "argv", ["bench.py", "--protein", "GFP", "--workload", "potts+cnn", "--gpus", "8"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0 and "--nproc-per-node=8" in seen["cmd"]
    assert seen["cmd"][-6:] == ["--protein", "GFP", "--workload", "potts+cnn", "--gpus", "8"]
    for protein, lam in (("PABP", 5.0), ("UBE4B", 0.5), ("GFP", 15.0)):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--protein", protein, "--workload", "potts+cnn"])
        assert bench.resolve_defaults(bench.parse()).lamda == lam
    monkeypatch.setattr(sys, "argv", ["bench.py", "--protein", "GFP", "--workload", "potts+cnn", "--lamda", "2.5"])
    assert bench.resolve_defaults(bench.parse()).lamda == 2.5
    monkeypatch.undo()                                               # (bench.subprocess IS this module's subprocess)
    # as a rank: WORLD_SIZE must equal --gpus
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_host_layer_under_address_sanitizer():
    """The host side of the C ABI (no device code) against a mock HIP runtime under ASan + LSan: the whole call
    sequence with all three experts, then once more per fallible runtime call with that call failing, so that every
    clean-up path runs (tests/hostcheck/). Any leak or out-of-bounds copy fails the run."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "build_and_run.sh")
    r = subprocess.run(["bash", script, "sweep"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    m = re.search(r"hostcheck ok: (\d+) fallible runtime calls per sequence, (\d+) injected failures handled", r.stdout)
    assert m and int(m.group(1)) > 100 and m.group(1) == m.group(2), r.stdout
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
    # the same script checks the split-precision helpers of cnn.h on the host (tests/hostcheck/splitcheck.cpp): the power-of-two
    # scales, the two-term fp16 split of an operand (within 2^-22) and a product from its three cross terms (within 3 * 2^-22)
    sp = re.search(r"splitcheck ok: split within ([0-9.]+) x 2\^-22, product within ([0-9.]+) x 2\^-22", r.stdout)
    assert sp and float(sp.group(1)) <= 1.0 and float(sp.group(2)) <= 3.0, r.stdout[-2000:]
