"""TEST INFRASTRUCTURE ONLY - the yardstick for the 16-bit MoVQ decode: how far is the REFERENCE'S OWN half-precision decode from its fp32
decode?  Under use_fp16 the reference runs `self.image_encoder = self.image_encoder.half()` and decodes half latents
(kandinsky2/kandinsky2_1_model.py:92-94, 287-288).  This script imports the reference's MOVQ module (build container only), decodes the golden
fixtures' latents (same seeded weights / latent as tests/golden/movq_256px.pt, movq_768px.pt) in fp32 and with .half() on the CPU, and writes
the distances to tests/golden/ref_movq_half_drift.json; tests/test_movq_gpu.py asserts that the fp16 ENGINE is at least as close to the fp32
reference image as the reference's own half mode.

    python oracle/ref_movq_half_drift.py [--lats 32,96]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kandinsky2_amd as k22  # noqa: E402
from oracle import movq_ref, ref_loader  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lats", default="32,96")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cfg = k22.MOVQ_CONFIG_2_1
    arch = k22.MoVQArch(cfg["ddconfig"], cfg["embed_dim"])
    sd = k22.init_movq_state_dict(arch, seed=0)
    ae = ref_loader.ref("vqgan.autoencoder")
    out = {}
    for lat in [int(v) for v in a.lats.split(",")]:
        m = ae.MOVQ(ddconfig=cfg["ddconfig"], n_embed=cfg["n_embed"], embed_dim=cfg["embed_dim"]).eval()
        m.load_state_dict(sd, strict=False)
        g = torch.Generator().manual_seed(5)          # make_golden.movq_case: seed_z = 5
        z = torch.randn(1, 4, lat, lat, generator=g)
        with torch.no_grad():
            ref = m.decode(z)
            half = m.half().decode(z.half()).float()
        du = (movq_ref.process_images_u8(half).int() - movq_ref.process_images_u8(ref).int()).abs()
        r = {"px": 8 * lat, "scale": ref.abs().max().item(), "max_abs": (half - ref).abs().max().item(),
             "rel": ((half - ref).abs().max() / ref.abs().max()).item(), "uint8_max_diff": int(du.max().item()),
             "uint8_frac_differ": (du > 0).float().mean().item(), "uint8_frac_more_than_one": (du > 1).float().mean().item()}
        out[f"movq_{8 * lat}px"] = r
        print(f"reference MOVQ.half() vs its fp32 decode at {8 * lat} px: {r['rel']:.3e} of scale {r['scale']:.3f}; uint8 max diff {r['uint8_max_diff']}, "
              f"{100 * r['uint8_frac_differ']:.3f} % of the bytes differ, {100 * r['uint8_frac_more_than_one']:.4f} % by more than one level", flush=True)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_movq_half_drift.json")
    json.dump({"what": "reference MOVQ module .half() decode vs its fp32 decode, CPU, seeded weights (seed 0) and latent (seed 5) of the movq_*px fixtures",
               "cases": out}, open(dst, "w"), indent=1)
    print("->", dst)
