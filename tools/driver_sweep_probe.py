#!/usr/bin/env python3
"""The run_recbole-style entry (mi_oov.driver.run) over plugins x embedding sizes x bucket counts on the tests' toy dataset:
one short training + the uni-250 evaluation each; prints ok / FAIL per configuration.  Developer probe, GPU box."""
import itertools
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from test_driver import write_dataset  # noqa: E402

import mi_oov  # noqa: E402,F401
from mi_oov import driver  # noqa: E402

tmp = tempfile.mkdtemp()
os.chdir(tmp)
root = write_dataset(tmp)
plugins = (["--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov"], ["--inductive_embedder=slsh", "--add_oov_buckets", "--train_oov"],
           ["--inductive_embedder=dhe", "--dhe_num_hashes=32", "--train_oov"], ["--inductive_embedder=fdhe", "--dhe_num_hashes=8", "--dhe_layer_size=40", "--train_oov"],
           ["--inductive_embedder=dnn", "--dhe_layer_size=40", "--train_oov"], ["--inductive_embedder=knn"], ["--inductive_embedder=mean"],
           ["--inductive_embedder=zero"], ["--inductive_mapper=random", "--add_oov_buckets"],
           ["--inductive_mapper=random", "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov"])
n_ok = 0
for flags, D, nb, model in itertools.product(plugins, (16, 50, 64, 130, 300), (1, 8, 100), ("BPR", "DirectAU")):
    if model == "DirectAU" and (D not in (50, 64) or nb != 8):
        continue
    name = f"{model} {' '.join(f for f in flags if 'embedder' in f or 'mapper' in f)} D={D} buckets={nb}"
    try:
        args = driver.custom_parse_args(["x", f"--data_path={root}", "--dataset=toy", f"--model={model}", f"--embedding_size={D}",
                                         f"--user_oov_buckets={nb}", f"--item_oov_buckets={nb}", "--epochs=1", "--learning_rate=0.01",
                                         "--train_batch_size=2048"] + flags + (["--gamma=0.5"] if model == "DirectAU" else []))
        results, _ = driver.run(args)
        bad = [(s, k, v) for s, d in results.items() if isinstance(d, dict) for k, v in d.items() if not (np.isfinite(v) and 0.0 <= v <= 1.0)]
        if bad:
            print(f"BAD  {name}: {bad[:3]}", flush=True)
        else:
            n_ok += 1
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:200]}", flush=True)
print(f"{n_ok} configurations ok", flush=True)
