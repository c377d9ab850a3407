"""run_recbole-style entry over the training / evaluation flags of the reference (masking, freezing, epoch layout, mapper, batch sizes,
metrics) on the tests' toy datasets, transductive and pre-split inductive: one short run each.  Developer probe, GPU box."""
import itertools, os, sys, tempfile
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_driver import write_dataset, write_split_dataset
import mi_oov
from mi_oov import driver
tmp = tempfile.mkdtemp(); os.chdir(tmp)
root = write_dataset(tmp); write_split_dataset(tmp)
base = ["x", f"--data_path={root}", "--model=BPR", "--embedding_size=64", "--user_oov_buckets=8", "--item_oov_buckets=8", "--epochs=2",
        "--learning_rate=0.01", "--train_batch_size=1024", "--inductive_embedder=lsh", "--add_oov_buckets", "--train_oov"]
variants = [["--oov_feature_mask_rate=0.3"], ["--oov_freeze_embedding=True"], ["--oov_freeze_embedding=True", "--oov_freeze_skip_optim=True"],
            ["--oov_only_epoch=False"], ["--oov_train_ratio=0.5"], ["--oov_shuffle_epoch=False"], ["--oov_normalization_type=global"],
            ["--oov_normalization_type=none"], ["--oov_hash_function=fast", "--inductive_mapper=random"], ["--oov_eval_batch_size=4096"],
            ["--eval_batch_size=512"], ["--weight_decay=0.001"], ["--model_eval_type=ranking"], ["--inductive_eval=False"],
            ["--oov_debug_skip_train=True"], ["--oov_debug_skip_eval=True"], ["--topk=[5,10,50]"], ["--metrics=[Recall,NDCG,MRR,Hit,Precision,MAP]"]]
ok = 0
for ds, v in itertools.product((["--dataset=toy"], ["--dataset=toy_ind", "--benchmark_filename=train,empty,test_filt"]), variants):
    name = " ".join(ds + v)
    try:
        results, _ = driver.run(driver.custom_parse_args(base + ds + v))
        ok += 1
    except Exception as e:
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:220]}", flush=True)
print(ok, "ok of", 2 * len(variants))
