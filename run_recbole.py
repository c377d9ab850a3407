#!/usr/bin/env python3
"""Entry point with the reference's command line (src/run_recbole.py:202-266):

    python run_recbole.py --dataset=ml-100k --data_path=/path/to/dataset --model=BPR \\
        --inductive_embedder=lsh --add_oov_buckets --user_oov_buckets=8 --item_oov_buckets=8 \\
        --embedding_size=64 --train_oov --inductive_eval

Flags are parsed like the reference (`--key=value` typed, bare `--key` = True); see
improving-inductive-oov-recsys_amd/driver.py for what is (deliberately) simplified around the kernels.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import mi_oov  # noqa: E402,F401
from mi_oov import driver  # noqa: E402

if __name__ == "__main__":
    driver.run(driver.custom_parse_args())
