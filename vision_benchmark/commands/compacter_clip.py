#!/usr/bin/env python3
"""Launcher at the path the reference's scripts use: scripts/compacter_clip.sh does
`cd ../vision_benchmark; python commands/compacter_clip.py --ds resources/datasets/<d>.yaml --model resources/model/<m>.yaml ...`
(reference: vision_benchmark/commands/compacter_clip.py).  Same flags and outputs; the work is pevit_amd.commands.compacter_clip."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from pevit_amd.commands.compacter_clip import main  # noqa: E402

if __name__ == "__main__":
    main()
