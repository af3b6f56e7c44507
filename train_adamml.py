#!/usr/bin/env python3
"""`python3 train_adamml.py ...` with the reference's README command lines, on the MI355X HIP path (adamml_amd/train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    from adamml_amd.train import main
    main()
