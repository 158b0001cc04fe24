#!/usr/bin/env python
"""Emit dpp_blocks_f64_gen.hpp: the float64 form of gen_dpp_blocks.py's blocks (`v_fmac_f64_dpp ... row_newbcast:k`,
`RiccatiBlocks64<NX, NU, 16>`) for the float64 row kernels (f64_row_kernels.hpp)."""
import gen_dpp_blocks

if __name__ == "__main__":
    gen_dpp_blocks.main("f64")
