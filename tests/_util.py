"""Shared helpers for the GPU parity tests."""
import torch


def rel_err(a, b):
    """max |a-b| / max |b|  (scale-relative max error; the metric the tolerances below are written in)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    fin = torch.isfinite(b)
    if not fin.all():
        assert torch.equal(torch.isfinite(a), fin), "non-finite pattern differs"
        a, b = a[fin], b[fin]
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# north_star tolerances: 1e-3 rel fp32 / 1e-2 bf16 on forward outputs; gradients in bf16 get 3e-2
TOL = {torch.float32: (1e-3, 1e-3), torch.bfloat16: (1e-2, 3e-2)}


def cell_dims_from_sd(sd, meta, enc_dim):
    """Recover the SummaryMixing constructor arguments from a reference state_dict."""
    def blocks(prefix):
        dims, i = [], 0
        while True:
            n = "linear" if i == 0 else f"linear_{i - 1}"
            if f"{prefix}.{n}.w.weight" in sd:
                dims.append(sd[f"{prefix}.{n}.w.weight"].shape[0])
            elif f"{prefix}.{n}.weights" in sd:
                w = sd[f"{prefix}.{n}.weights"]
                dims.append(w.shape[0] * w.shape[2])
            else:
                return dims
            i += 1
    kw = dict(enc_dim=enc_dim, nhead=meta["nhead"], mode=meta["mode"], local_proj_out_dim=meta["local_proj_out_dim"])
    lp, sp, mg = blocks("local_proj"), blocks("summary_proj"), blocks("summary_local_merging")
    kw["local_proj_hid_dim"] = lp[:-1] if lp else [meta["local_proj_out_dim"]]
    if sp:
        kw["summary_hid_dim"], kw["summary_out_dim"] = sp[:-1], sp[-1]
    else:
        kw["summary_hid_dim"], kw["summary_out_dim"] = [8], mg[-1]
    return kw


def rms_rel(a, b):
    """||a-b||_2 / ||b||_2 over the finite entries (the RMS-relative error reported next to rel_err)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    fin = torch.isfinite(b)
    a, b = a[fin], b[fin]
    if b.numel() == 0:
        return 0.0
    return float((a - b).norm() / (b.norm() + 1e-300))


def report(name, entries):
    """Append measured errors to gpurun_out/parity_errors.jsonl (copied into DESIGN.md §I.1 / profiles/ by hand)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_errors.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **entries}) + "\n")


def autocast_reference_layer(name):
    """The REFERENCE's own bf16 mode on a layer golden: the oracle layer under torch.autocast(bfloat16) on the CPU (Linear / conv
    in bf16, LayerNorm / sums in float32 - the recipes' `precision: bf16`).  -> dict(y, gx, grads: the autocast results as float32
    tensors; floor = (forward, dL/dx, worst parameter-gradient) max-rel error of those against the float32 golden).  Used to
    REPORT how far a bf16 implementation can be expected to sit from float32 and for the DIRECT bf16-vs-bf16 comparison; the
    asserted bars of the tests are fixed numbers."""
    import torch
    from oracle import smx_oracle as O
    from tests import _golden as G
    meta, a, sd, grads = G.load(name)
    sdp = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    x = a["x"].float().clone().requires_grad_(True)
    fn = O.conformer_layer if meta["kind"] == "conformer_layer" else O.branchformer_layer
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
        y = fn(x, sdp, "", meta["act"], meta["mode"], meta["local_proj_out_dim"], None, a["pad_mask"])
    (y.float() * a["r"]).sum().backward()
    pg = {k: sdp[k].grad.detach().clone() for k in grads if sdp[k].grad is not None}
    perr = max(rel_err(pg[k], g) for k, g in grads.items() if k in pg)
    return {"y": y.detach().float(), "gx": x.grad.detach().clone(), "grads": pg,
            "floor": (rel_err(y.float(), a["y"]), rel_err(x.grad, a["gx"]), perr)}


def autocast_floor_layer(name):
    """(forward, dL/dx, worst parameter-gradient) max-rel error of the reference's own bf16 autocast against the float32 golden."""
    return autocast_reference_layer(name)["floor"]
