"""BC7 blocks and the texels an independent decoder (Pillow's DDS reader) gives for them
-> tests/golden/bc7_blocks.npz:  blocks (n, 16) uint8, texels (n, 16, 4) uint8 (row-major inside a block).

  python tests/golden/make_bc7_blocks.py

Random bits under each of the eight mode prefixes (so every mode, partition, rotation, index
selector and p-bit combination turns up), blocks of plain random bytes, and extreme end points.  The
reserved mode (first byte zero) is left out: the format defines it as all zeros, which a test states
by itself, and software decoders disagree about it.
"""
import os

import numpy as np

from make_bc7_tables import decode_blocks

rng = np.random.default_rng(7)
blocks = []
for mode in range(8):
    b = rng.integers(0, 256, (160, 16), dtype=np.uint8)
    b[:, 0] = (b[:, 0] & ~np.uint8((1 << (mode + 1)) - 1)) | np.uint8(1 << mode)
    blocks.append(b)
    for fill in (0x00, 0xFF):  # every field after the mode all zeros / all ones
        e = np.full((1, 16), fill, np.uint8)
        e[0, 0] = (fill & ~((1 << (mode + 1)) - 1) & 0xFF) | (1 << mode)
        blocks.append(e)
blocks.append(rng.integers(0, 256, (192, 16), dtype=np.uint8))
blocks = np.concatenate(blocks)
blocks = blocks[blocks[:, 0] != 0]
texels = decode_blocks(blocks)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bc7_blocks.npz")
np.savez_compressed(out, blocks=blocks, texels=texels)
print(out, blocks.shape, texels.shape)
