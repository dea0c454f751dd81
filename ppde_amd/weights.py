"""Readers for the reference's weight files (formats listed in SURVEY.md §8(b)).

  potts.pkl                      ppde/nets.py:247-262   dict J_ij, h_i, index_list, reg_coef
  onehot_cnn_seed={0,1,2}.pt     ppde/nets.py:423       {'model': state_dict of OnehotCNN}
  results-...-seed=K-linear.pkl  ppde/nets.py:325-329   dict coef_, intercept_, reg_coef
  wt.fasta                       ppde/nets.py:255-260   id 'NAME/first-last' -> offset of index_list
"""
import os
import pickle

import numpy as np
import torch

from .encoding import read_fasta, seqs_to_idx


class PottsParams:
    """Couplings, fields and the 0-based window of a potts.pkl next to its wt.fasta."""

    def __init__(self, protein_dir):
        with open(os.path.join(protein_dir, "potts.pkl"), "rb") as fh:
            d = pickle.load(fh)
        self.J = np.ascontiguousarray(np.asarray(d["J_ij"], dtype=np.float32))
        self.h = np.ascontiguousarray(np.asarray(d["h_i"], dtype=np.float32))
        self.reg_coef = d["reg_coef"]
        self.wtseqs, ids = read_fasta(os.path.join(protein_dir, "wt.fasta"), return_ids=True)
        self.offset = int(ids[0].split("/")[-1].split("-")[0]) if "/" in ids[0] else 1
        self.index_list = np.asarray(d["index_list"]).astype(np.int64) - self.offset
        self.seq_len = int(self.index_list.shape[0])
        if self.J.shape != (self.seq_len, self.seq_len, 20, 20) or self.h.shape != (self.seq_len, 20):
            raise ValueError(f"potts.pkl shapes {self.J.shape}/{self.h.shape} do not match index_list of length {self.seq_len}")
        if not np.array_equal(self.index_list, np.arange(self.index_list[0], self.index_list[0] + self.seq_len)):
            # the reference slices x[:, index_list[0]:index_list[-1]+1] and reshapes to [.., seq_len, 20] (nets.py:280,285)
            raise ValueError("potts.pkl index_list must be contiguous")
        self.win_start = int(self.index_list[0])


def load_cnn_states(protein_dir, seeds=(0, 1, 2)):
    """List of numpy state dicts of the OnehotCNN checkpoints."""
    out = []
    for s in seeds:
        ck = torch.load(os.path.join(protein_dir, f"onehot_cnn_seed={s}.pt"), map_location="cpu")
        out.append({k: v.detach().cpu().numpy().astype(np.float32) for k, v in ck["model"].items()})
    return out


def load_linear(protein_dir, seeds=range(20)):
    out = []
    for s in seeds:
        with open(os.path.join(protein_dir, f"results-predictor=ev+onehot-train=-1-seed={s}-linear.pkl"), "rb") as fh:
            d = pickle.load(fh)
        out.append((np.asarray(d["coef_"], dtype=np.float32), float(np.asarray(d["intercept_"]).reshape(-1)[0]), float(d["reg_coef"])))
    return out


def load_wt(protein_dir):
    seqs = read_fasta(os.path.join(protein_dir, "wt.fasta"), return_ids=False)
    return seqs, seqs_to_idx(seqs)


def load_esm2_state(path, with_heads=False):
    """ESM-2 checkpoint -> state dict with the module's own parameter names (embed_tokens.weight, layers.i..., lm_head...).
    Accepts the published files ({'cfg': {'model': Namespace}, 'model': {...}} with 'encoder.sentence_encoder.' /
    'encoder.' prefixes, which facebookresearch/esm strips on load) as well as already stripped dicts.
    with_heads: also return cfg.model.encoder_attention_heads (None when the file has no cfg)."""
    import argparse
    # the published checkpoints pickle an argparse.Namespace next to the tensors: allow exactly that class under the
    # weights-only unpickler instead of switching it off for a downloaded file
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    heads = None
    cfg = ck.get("cfg") if isinstance(ck, dict) else None
    if isinstance(cfg, dict) and "model" in cfg:
        heads = getattr(cfg["model"], "encoder_attention_heads", None)
    out = {}
    for k, v in sd.items():
        for pre in ("encoder.sentence_encoder.", "encoder."):
            if k.startswith(pre):
                k = k[len(pre):]
                break
        if k == "lm_head.weight" or k.endswith("inv_freq") or "contact_head" in k:
            continue                                    # tied to embed_tokens / buffers / unused head
        out[k] = v.detach().cpu().numpy().astype(np.float32)
    return (out, heads) if with_heads else out
