"""Golden vectors produced by EXECUTING THE REFERENCE'S SHIPPED SPIR-V (shader/spv/*.comp.spv)
with oracle/spirv_interp.py on the reference's own inputs (data/spectrum.bin, data/omega.bin).

Run in the build container only (it reads /root/reference, which does not travel):

    python tests/golden/make_spirv_golden.py

Outputs tests/golden/spirv_frame512_t{0,1,10}.npz (data only): 64x64 crop and an every-8th-texel
subsample of the RGBA32F displacement image, probe texels, per-channel aggregates, and CRC-32 of
the raw bytes of the image and of every intermediate buffer after each dispatch group
(propagate, fft_row x3, fft_col x3) in the order height, disp_x, disp_z.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ocean_oracle as oc  # noqa: E402
from oracle import spirv_interp as si  # noqa: E402

SPV = "/root/reference/shader/spv"


def main():
    h0, om = oc.load_reference_inputs(os.path.join(HERE, "spectrum.bin"), os.path.join(HERE, "omega.bin"))
    for t in (0, 1, 10):
        img, st = si.run_reference_frame(SPV, h0, om, float(t), return_stages=True)
        ch = img[..., :3].astype(np.float64)
        crcs = {f"crc_{k}": np.array([zlib.crc32(np.ascontiguousarray(b).tobytes()) for b in v], np.uint32)
                for k, v in st.items()}
        np.savez_compressed(
            os.path.join(HERE, f"spirv_frame512_t{t}.npz"),
            crop=img[:64, :64].copy(), sub8=img[::8, ::8].copy(),
            crc_image=np.uint32(zlib.crc32(img.tobytes())),
            sum=ch.sum((0, 1)), l2=np.sqrt((ch ** 2).sum((0, 1))), max=np.abs(ch).max((0, 1)), **crcs)
        print(t, "crc", zlib.crc32(img.tobytes()))


if __name__ == "__main__":
    main()
