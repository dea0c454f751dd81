"""Seeded synthetic weights in the reference's on-disk formats.

The reference's `potts.pkl` files are missing blobs (`.MISSING_LARGE_BLOBS`) and the
GPU box has no copy of the reference at all, so every test and benchmark of the path runs on
synthetic parameters generated here:

  potts.pkl                     dict J_ij [L',L',20,20], h_i [L',20], index_list [L'], reg_coef
                                (schema read at ppde/nets.py:247-251)
  onehot_cnn_seed={0,1,2}.pt    {'model': state_dict of OnehotCNN(20, 5, L)}  (ppde/nets.py:350-376, :423)
  results-predictor=ev+onehot-train=-1-seed={0..19}-linear.pkl
                                dict coef_ [1+L*20], intercept_, reg_coef     (ppde/nets.py:325-329)
  wt.fasta                      one record; id "<name>/<first>-<last>" gives the Potts offset (ppde/nets.py:257-260)

The wild-type sequences below are the public assay wild types (data, 96/104/237 residues).
"""
import os
import pickle
import numpy as np

A = 20

PROTEINS = {
    # name: (fasta id, wild-type sequence, default Potts window (start, length) used by the survey's probes)
    "PABP_YEAST_Fields2013": (
        "PABP_YEAST/115-210",
        "QRDPSLRKKGSGNIFIKNLHPDIDNKALYDTFSVFGDILSSKIATDENGKSKGFGFVHFEEEGAAKEAIDALNGMLLNGQEIYVAPHLSRKERDSQ",
        (8, 80)),
    "UBE4B_MOUSE_Klevit2013-nscor_log2_ratio": (
        "UBE4B_MOUSE/1070-1173",
        "IAIEKFKLLAEKVEEIVAKNARAEIDYSDAPDEFRDPLMDTLMTDPVRLPSGTVMDRSIILRHLLNSPTDPFNRQMLTESMLEPVPELKEQIQAWMREKQSSDH",
        (23, 76)),
    "GFP_AEQVI_Sarkisyan2016": (
        "sarkisyan_wt",
        "SKGEELFTGVVPILVELDGDVNGHKFSVSGEGEGDATYGKLTLKFICTTGKLPVPWPTLVTTLSYGVQCFSRYPDHMKQHDFFKSAMPEGYVQERTIFFKDDGNYKTRAEV"
        "KFEGDTLVNRIELKGIDFKEDGNILGHKLEYNYNSHNVYIMADKQKNGIKVNFKIRHNIEDGSVQLADHYQQNTPIGDGPVLLPDNHYLSTQSALSKDPNEKRDHMVLLEF"
        "VTAAGITHGMDELYK",
        (0, 237)),
    # toy protein for fast tests
    "TOY24": ("TOY24/11-34", "MKTAYIAKQRQISFVKSHFSRQLE", (4, 16)),
}


def fasta_offset(fasta_id):
    """First residue number from an id like 'PABP_YEAST/115-210'; 1 when the id has no range (nets.py:257-260)."""
    if "/" in fasta_id:
        return int(fasta_id.split("/")[-1].split("-")[0])
    return 1


def make_potts(Lp, seed=1234, sigma_J=0.05, sigma_h=0.5, symmetric=True):
    """Seeded couplings/fields. symmetric=True gives J[i,j,k,l] == J[j,i,l,k] with zero diagonal blocks."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((Lp, Lp, A, A), dtype=np.float32) * np.float32(sigma_J)
    if symmetric:
        J = np.float32(0.5) * (G + G.transpose(1, 0, 3, 2))
    else:
        J = G
    J[np.arange(Lp), np.arange(Lp)] = 0.0
    h = rng.standard_normal((Lp, A), dtype=np.float32) * np.float32(sigma_h)
    return np.ascontiguousarray(J, dtype=np.float32), h.astype(np.float32)


def make_cnn_state(L, seed, kernel_size=5):
    """numpy state dict with OnehotCNN(20, 5, L)'s parameter names/shapes, U(-1/sqrt(fan_in), +) like nn defaults."""
    rng = np.random.default_rng(10_000 + seed)

    def u(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-b, b, size=shape).astype(np.float32)

    return {
        "encoder.weight": u((L, A, kernel_size), A * kernel_size),
        "encoder.bias": u((L,), A * kernel_size),
        "embedding.0.weight": u((2 * L, L), L),
        "embedding.0.bias": u((2 * L,), L),
        "decoder.weight": u((1, 2 * L), 2 * L),
        "decoder.bias": u((1,), 2 * L),
    }


ESM_VOCAB = 33


def make_esm2_state(n_layers, dim, heads, ffn, seed=0):
    """Seeded random weights with ESM-2's parameter names and shapes (facebookresearch/esm `ESM2`): the real
    checkpoints (esm2_t30_150M_UR50D = 30 layers, 640 wide, 20 heads, 2560 ffn) come from torch hub at run time in the
    reference and are not available offline. Scales follow the usual transformer initialisation (0.02-sigma normal
    matrices, unit layer-norm gains) with the embedding made larger so that the logits have a spread."""
    rng = np.random.default_rng(30_000 + seed)
    nrm = lambda *shape, s=0.02: (rng.standard_normal(shape) * s).astype(np.float32)
    st = {"embed_tokens.weight": nrm(ESM_VOCAB, dim, s=0.1)}
    for i in range(n_layers):
        pre = f"layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            st[pre + f"self_attn.{nm}.weight"] = nrm(dim, dim, s=1.0 / np.sqrt(dim))
            st[pre + f"self_attn.{nm}.bias"] = nrm(dim)
        for nm in ("self_attn_layer_norm", "final_layer_norm"):
            st[pre + nm + ".weight"] = (1.0 + nrm(dim, s=0.05)).astype(np.float32)
            st[pre + nm + ".bias"] = nrm(dim)
        st[pre + "fc1.weight"] = nrm(ffn, dim, s=1.0 / np.sqrt(dim))
        st[pre + "fc1.bias"] = nrm(ffn)
        st[pre + "fc2.weight"] = nrm(dim, ffn, s=1.0 / np.sqrt(ffn))
        st[pre + "fc2.bias"] = nrm(dim)
    st["emb_layer_norm_after.weight"] = (1.0 + nrm(dim, s=0.05)).astype(np.float32)
    st["emb_layer_norm_after.bias"] = nrm(dim)
    st["lm_head.dense.weight"] = nrm(dim, dim, s=1.0 / np.sqrt(dim))
    st["lm_head.dense.bias"] = nrm(dim)
    st["lm_head.layer_norm.weight"] = (1.0 + nrm(dim, s=0.05)).astype(np.float32)
    st["lm_head.layer_norm.bias"] = nrm(dim)
    st["lm_head.bias"] = nrm(ESM_VOCAB)
    return st


def write_esm2_checkpoint(path, n_layers, dim, heads, ffn, seed=0):
    """Synthetic ESM-2 checkpoint in the published file format (prefixed names, tied lm_head.weight included)."""
    import torch
    st = make_esm2_state(n_layers, dim, heads, ffn, seed)
    sd = {}
    for k, v in st.items():
        pre = "encoder." if k.startswith("lm_head") else "encoder.sentence_encoder."
        sd[pre + k] = torch.from_numpy(v)
    sd["encoder.lm_head.weight"] = sd["encoder.sentence_encoder.embed_tokens.weight"]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    # the published files carry the architecture as an argparse.Namespace under cfg.model (facebookresearch/esm reads
    # cfg.model.encoder_layers / encoder_embed_dim / encoder_attention_heads / token_dropout from it)
    import argparse
    cfg = argparse.Namespace(encoder_layers=n_layers, encoder_embed_dim=dim, encoder_attention_heads=heads, token_dropout=True,
                             arch="roberta_large")
    torch.save({"cfg": {"model": cfg}, "model": sd}, path)
    return st


def make_linear(L, seed):
    rng = np.random.default_rng(20_000 + seed)
    return {
        "coef_": (rng.standard_normal(1 + L * A) * 0.05).astype(np.float64),
        "intercept_": float(rng.standard_normal() * 0.1),
        "reg_coef": 1.0,
    }


def write_weights_dir(root, protein="PABP_YEAST_Fields2013", window=None, potts_seed=1234,
                      symmetric=True, cnn_seeds=(0, 1, 2), linear_seeds=range(20), wt_seq=None, fasta_id=None):
    """Create `<root>/<protein>/` holding synthetic files in the reference's formats. Returns the directory."""
    import torch
    fid, seq, win = PROTEINS[protein] if protein in PROTEINS else (fasta_id, wt_seq, window)
    if wt_seq is not None:
        seq = wt_seq
    if fasta_id is not None:
        fid = fasta_id
    if window is not None:
        win = window
    start, Lp = win
    L = len(seq)
    if start < 0 or start + Lp > L:
        raise ValueError(f"Potts window [{start}, {start + Lp}) does not fit a length-{L} wild type")
    d = os.path.join(root, protein)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "wt.fasta"), "w") as fh:
        fh.write(f">{fid}\n{seq}\n")
    J, h = make_potts(Lp, seed=potts_seed, symmetric=symmetric)
    with open(os.path.join(d, "potts.pkl"), "wb") as fh:
        pickle.dump({"J_ij": J, "h_i": h,
                     "index_list": np.arange(start, start + Lp, dtype=np.int64) + fasta_offset(fid),
                     "reg_coef": 1.0}, fh)
    for s in cnn_seeds:
        sd = {k: torch.from_numpy(v) for k, v in make_cnn_state(L, s).items()}
        torch.save({"model": sd}, os.path.join(d, f"onehot_cnn_seed={s}.pt"))
    for s in linear_seeds:
        with open(os.path.join(d, f"results-predictor=ev+onehot-train=-1-seed={s}-linear.pkl"), "wb") as fh:
            pickle.dump(make_linear(L, s), fh)
    return d
