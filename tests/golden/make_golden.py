"""Generates the committed golden fixtures from the oracle (run in the build container).

    python tests/golden/make_golden.py

Outputs (small, data only):
  kat_survey.json     -- the survey-time known-answer table, copied verbatim;
                         it was produced by an independent restatement, not by this oracle.
  frame512_t{0,1,10}.npz -- fp64-oracle output on data/*.bin: 64x64 top-left crop (as f32),
                         six probe texels and per-channel sum / l2 / max aggregates.
  frame256_t1.npz     -- same for the N=256 centre crop (BASELINE config 1).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ocean_oracle as oc  # noqa: E402

KAT = {  # t -> [(x, y, dx, h, dz)]
    "0": [[0, 0, -3.082524, -1.326607, -0.321310], [1, 0, -3.137091, -1.535278, -0.398301],
          [0, 1, -3.094816, -1.110866, -0.267298], [255, 255, -1.401235, 2.301178, -3.969325],
          [257, 256, -1.084222, 2.196662, -4.099258], [511, 511, -3.017880, -1.183992, -0.466976]],
    "1": [[0, 0, -1.814249, -1.339759, -0.750584], [255, 255, -2.136859, 0.991178, -3.493580],
          [511, 511, -2.101657, -1.447329, -0.912272]],
    "10": [[0, 0, -2.740192, -3.627667, -2.908885], [255, 255, -0.332904, 0.897933, -0.410262],
           [511, 511, -3.091887, -3.412703, -3.316658]],
    "aggregates": {
        "0": {"sum": [-165.5564, -389.5465, -165.5564], "l2": [1169.162, 1520.020, 1289.944],
              "max": [8.085795, 11.037553, 9.154893]},
        "1": {"sum": [-127.0227, -417.4863, -127.0227], "l2": [1160.214, 1523.330, 1293.455]},
        "10": {"sum": [232.1621, -314.2703, 232.1621], "l2": [1074.272, 1607.369, 1274.172]},
    },
}


def dump(name, out64):
    ch = out64[..., :3]
    np.savez_compressed(
        os.path.join(HERE, name),
        crop=ch[:64, :64].astype(np.float32),
        probes_xy=np.array([[0, 0], [1, 0], [0, 1], [out64.shape[0] // 2 - 1] * 2,
                            [out64.shape[0] // 2 + 1, out64.shape[0] // 2], [out64.shape[0] - 1] * 2]),
        probes=np.array([ch[y, x] for x, y in [(0, 0), (1, 0), (0, 1), (out64.shape[0] // 2 - 1,) * 2,
                                               (out64.shape[0] // 2 + 1, out64.shape[0] // 2),
                                               (out64.shape[0] - 1,) * 2]]),
        sum=ch.sum((0, 1)), l2=np.sqrt((ch ** 2).sum((0, 1))), max=np.abs(ch).max((0, 1)))


def main():
    with open(os.path.join(HERE, "kat_survey.json"), "w") as f:
        json.dump(KAT, f, indent=1)
    h0, om = oc.load_reference_inputs(os.path.join(HERE, "spectrum.bin"), os.path.join(HERE, "omega.bin"))
    for t in (0, 1, 10):
        dump(f"frame512_t{t}.npz", oc.frame_f64(h0, om, float(t)))
    dump("frame256_t1.npz", oc.frame_f64(oc.centre_crop(h0, 256), oc.centre_crop(om, 256), 1.0))


if __name__ == "__main__":
    main()
