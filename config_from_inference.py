#!/usr/bin/env python3
"""Extracts the configuration file from a slim inference checkpoint (same flags as the reference tool)."""
import argparse
import sys
from pathlib import Path

import k_diffusion_amd as K


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("checkpoint", type=Path, help="the inference checkpoint to extract the configuration from")
    p.add_argument("--output", "-o", type=Path, help="the output configuration file")
    args = p.parse_args(argv)
    text = K.checkpoint.read_config(args.checkpoint)
    out = args.output or args.checkpoint.with_suffix(".json")
    out.write_text(text)
    print(f"Saved configuration to {out}", file=sys.stderr)


if __name__ == "__main__":
    main()
