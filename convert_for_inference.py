#!/usr/bin/env python3
"""Converts a k-diffusion training checkpoint (.pth) to a slim inference checkpoint (safetensors + config metadata).
Same flags as the reference tool; the work is in k-diffusion_amd/checkpoint.py."""
import argparse
import json
import sys
from pathlib import Path

import k_diffusion_amd as K


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("checkpoint", type=Path, help="the training checkpoint to convert")
    p.add_argument("--config", type=Path, help="override the checkpoint's configuration")
    p.add_argument("--output", "-o", type=Path, help="the output slim checkpoint")
    p.add_argument("--dtype", type=str, choices=sorted(K.checkpoint.DTYPES), default="fp16", help="the output dtype")
    p.add_argument("--unsafe", action="store_true", help="fall back to the full unpickler for checkpoints that need it (trusted files only)")
    args = p.parse_args(argv)
    override = json.loads(args.config.read_text()) if args.config else None
    print(f"Loading training checkpoint {args.checkpoint}...", file=sys.stderr)
    out = K.checkpoint.convert_training_checkpoint(args.checkpoint, args.output, override, args.dtype, unsafe=args.unsafe)
    print(f"Saved inference checkpoint to {out}", file=sys.stderr)


if __name__ == "__main__":
    main()
