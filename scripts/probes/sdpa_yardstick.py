#!/usr/bin/env python3
"""Yardstick only (not product code): torch's scaled_dot_product_attention (fp16, forward and forward + backward) at the
transformer expert's attention shape (256 chains x 20 heads x 104 residues x 32), against tf_attn_fwd / tf_attn_bwd's
48 / 156 us per layer (which also apply the rotary embedding and its transpose). python scripts/probes/sdpa_yardstick.py"""
import torch
torch.manual_seed(0)
B, H, L, D = 256, 20, 104, 32
q, k, v = [((torch.rand(B, H, L, D, device="cuda") - 0.5).half()).requires_grad_() for _ in range(3)]
do = (torch.rand(B, H, L, D, device="cuda") - 0.5).half()
def fwd():
    return torch.nn.functional.scaled_dot_product_attention(q, k, v)
def fwd_bwd():
    o = fwd()
    o.backward(do)
    q.grad = k.grad = v.grad = None
for name, f in (("forward", fwd), ("forward + backward", fwd_bwd)):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"torch SDPA {name:18s}: {e0.elapsed_time(e1) * 1e3 / 30:8.1f} us", flush=True)
