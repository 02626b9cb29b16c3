"""Hardware probe: device info, f64 MFMA D-register layout, sustained fp64 MFMA /
FMA rates, HBM write / copy bandwidth.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402


def main():
    e = Engine(0)
    out = {"device": e.info()}
    lay = e.probe_mfma_layout()
    guide = all(lay[l, r] == 16 * ((l >> 4) + 4 * r) + (l & 15) for l in range(64) for r in range(4))
    f32style = all(lay[l, r] == 16 * (4 * (l >> 4) + r) + (l & 15) for l in range(64) for r in range(4))
    out["mfma_f64_layout"] = {"row=(l>>4)+4r": guide, "row=4(l>>4)+r": f32style,
                              "lane0": lay[0].tolist(), "lane16": lay[16].tolist()}
    out["mfma_f64_tflops"] = [e.probe_mfma_f64() for _ in range(3)]
    out["fma_f64_tflops"] = [e.probe_fma_f64() for _ in range(3)]
    out["empty_kernel_chain_us_per_launch"] = [e.probe_launch(2000) for _ in range(3)]
    out["hbm_write_copy_gbs_1GiB"] = e.probe_hbm(1 << 30)
    out["hbm_write_copy_gbs_128MiB"] = e.probe_hbm(1 << 27)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
