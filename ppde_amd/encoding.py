"""Amino-acid encoding contract of the path.

Restates the alphabet order and one-hot conventions the reference uses
(ppde/third_party/hsu/data_utils.py:48-70 `aa_to_int`, :150-157 `seqs_to_onehot`,
:167-175 `onehot2seq`; ppde/third_party/hsu/io_utils.py:178-188 `read_fasta`).

The device-side state is the index form (`uint8 [n, L]`, one byte per residue);
the fp32 one-hot form only exists at the API edge.
"""
import numpy as np

ALPHABET = "ACDEFGHIKLMNPQRSTVWY"
VOCAB_SIZE = 20
aa_to_int = {a: i for i, a in enumerate(ALPHABET)}
int_to_aa = {i: a for a, i in aa_to_int.items()}


def seqs_to_idx(seqs):
    """list of equal-length strings -> uint8 [n, L] residue indices (0..19)."""
    if len(seqs) == 0:
        return np.zeros((0, 0), dtype=np.uint8)
    L = max(len(s) for s in seqs)
    out = np.zeros((len(seqs), L), dtype=np.uint8)  # short rows are 'A'-padded like the reference (pad value 0)
    for r, s in enumerate(seqs):
        try:
            out[r, :len(s)] = [aa_to_int[c] for c in s.strip()]
        except KeyError as e:
            raise KeyError(f"residue {e.args[0]!r} is not one of the 20 canonical amino acids") from None
    return out


def idx_to_onehot(idx, dtype=np.int64):
    """uint8 [n, L] -> one-hot [n, L, 20] (integer typed like the reference's seqs_to_onehot)."""
    idx = np.asarray(idx)
    out = np.zeros(idx.shape + (VOCAB_SIZE,), dtype=dtype)
    np.put_along_axis(out, idx[..., None].astype(np.int64), 1, axis=-1)
    return out


def seqs_to_onehot(seqs):
    """list of strings -> int one-hot [n, L, 20] (data_utils.py:150-157)."""
    return idx_to_onehot(seqs_to_idx(seqs))


def onehot_to_idx(onehots):
    """[n, L, 20] -> uint8 [n, L]; argmax per residue, first index on ties (np.argmax), as onehot2seq does."""
    return np.argmax(np.asarray(onehots), axis=-1).astype(np.uint8)


def idx_to_seqs(idx):
    return ["".join(int_to_aa[int(a)] for a in row) for row in np.asarray(idx)]


def onehot2seq(onehots):
    """[n, L, 20] -> list of strings (data_utils.py:167-175)."""
    return idx_to_seqs(onehot_to_idx(onehots))


def read_fasta(filename, return_ids=False):
    """Minimal FASTA reader with the reference's return contract (io_utils.py:178-188).

    The record id is the header up to the first whitespace (Biopython's `record.id`)."""
    seqs, ids = [], []
    with open(filename) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                ids.append(line[1:].split()[0] if len(line) > 1 else "")
                seqs.append("")
            else:
                if not seqs:
                    raise ValueError(f"{filename}: sequence data before the first '>' header")
                seqs[-1] += line
    if return_ids:
        return seqs, ids
    return seqs
