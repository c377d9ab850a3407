"""CPU: the C-ABI shared library loads and exports every symbol include/mi_oov.h declares.
No compute is launched here (there is no GPU in the build container)."""
import ctypes

import pytest

import __graft_entry__ as entry


@pytest.fixture(scope="module")
def lib():
    import mi_oov
    if not mi_oov.available():
        entry.build()
    return mi_oov._cabi.lib()


def test_every_header_symbol_is_exported(lib):
    syms = entry.header_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(__import__("mi_oov").LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/mi_oov.h but not exported"
    from mi_oov import _cabi
    assert set(_cabi.EXPORTS) == set(syms), "ctypes signature table out of sync with the header"


def test_version_and_strerror(lib):
    assert lib.mi_oov_version() == 100
    assert lib.mi_oov_strerror(0) == b"ok"
    for code in range(-7, 0):
        assert lib.mi_oov_strerror(code) != b"unknown error code"
    assert lib.mi_oov_strerror(-99) == b"unknown error code"


def test_argument_validation_without_gpu(lib):
    # shape/NULL validation happens before any HIP call, so it is safe on a GPU-less host
    assert lib.mi_oov_lsh_embed(None, 0, None, 10, 4, None, 2, None, 4, None, None, None) == 0  # empty batch
    assert lib.mi_oov_lsh_embed(None, 5, None, 10, 4, None, 2, None, 4, None, None, None) == -1  # NULL
    assert lib.mi_oov_lsh_embed(None, -1, None, 10, 4, None, 2, None, 4, None, None, None) == -2  # shape
    assert lib.mi_oov_mapper_map(None, 5, 9, 1, 1, None, None) == -3  # unknown hash kind
    assert lib.mi_oov_siphash24_mod(None, 5, None, 4, 1000, None, None) == -2  # mod not a power of two
    assert lib.mi_oov_col_mean_workspace(10_000_000, 64) == 1023 * 64  # <= 1024 row partitions of 9776 rows


def test_argument_validation_of_the_widened_entry_points(lib):
    """Error codes of the backward / evaluation / scoring entry points (all checked before any HIP call)."""
    assert lib.mi_oov_lsh_backward_workspace(65536, 8, 64) == 1024 * 8 * 64
    assert lib.mi_oov_lsh_backward_workspace(0, 8, 64) == 0
    assert lib.mi_oov_lsh_embed_backward(None, None, 5, 8, 300, None, None, None) == -2      # D > 256
    assert lib.mi_oov_lsh_embed_backward(None, None, 5, 8, 64, None, None, None) == -1       # NULL output
    assert lib.mi_oov_slsh_embed_backward(None, None, 5, 0, 64, None, None, None) == -2      # no buckets
    assert lib.mi_oov_scatter_add_rows(None, 0, None, 10, 64, None, None) == 0               # nothing to add
    assert lib.mi_oov_scatter_add_rows(None, 5, None, 10, 64, None, None) == -1
    assert lib.mi_oov_scatter_add_rows(None, 5, None, 10, 0, None, None) == -2
    assert lib.mi_oov_segment_topk(None, None, None, 0, 10, 0, 100, None, None, None) == 0   # no segments
    assert lib.mi_oov_segment_topk(None, None, None, 3, 257, 0, 100, None, None, None) == -2  # k > 256
    assert lib.mi_oov_segment_topk(None, None, None, 3, 10, 0, 100, None, None, None) == -1
    assert lib.mi_oov_topk_hits(None, 3, 0, None, None, None, None) == -2
    assert lib.mi_oov_topk_hits(None, 3, 10, None, None, None, None) == -1
    assert lib.mi_oov_score_topk(None, 4, None, 0, 64, 10, 0, None, None, None, None) == -2  # empty catalogue
    assert lib.mi_oov_score_topk(None, 4, None, 100, 64, 10, 0, None, None, None, None) == -1
    assert lib.mi_oov_score_topk_workspace(4096, 50_000, 20) > 0
    ws = lib.mi_oov_score_topk_masked_workspace(4096, 50_000, 64, 20)                        # lists + 4096 x 782 mask words
    assert ws >= lib.mi_oov_score_topk_workspace(4096, 50_000, 20) + 4096 * 782 * 8
    assert lib.mi_oov_score_topk_masked_workspace(4096, 50_000, 32, 20) == ws                # narrower rows: taken too
    assert lib.mi_oov_score_topk_masked_workspace(4096, 50_000, 129, 20) == 0                # D > 128: not taken
    assert lib.mi_oov_score_topk_masked_workspace(4096, 50_000, 65, 20) == ws                # (sized for two k-halves whatever D)
    assert lib.mi_oov_score_topk_masked_workspace(4096, 1000, 64, 20) == 0                   # catalogue < 128 k
    assert lib.mi_oov_score_topk_masked(None, 4, None, 50_000, 129, 20, 0, None, None, None, None, None, None) == -2
    assert lib.mi_oov_score_topk_masked(None, 4, None, 50_000, 64, 20, 0, None, None, None, None, None, None) == -1
    assert lib.mi_oov_score_topk_masked(None, 0, None, 50_000, 64, 20, 0, None, None, None, None, None, None) == 0
    assert lib.mi_oov_topk_catalogue_bytes(50_000, 64) >= 50_000 * 128 + 512 * 4            # bf16 rows + partial norm maxima
    assert lib.mi_oov_topk_catalogue_bytes(50_000, 32) == lib.mi_oov_topk_catalogue_bytes(50_000, 64) and lib.mi_oov_topk_catalogue_bytes(50_000, 129) == 0
    assert lib.mi_oov_topk_catalogue_bytes(50_000, 65) == lib.mi_oov_topk_catalogue_bytes(50_000, 128) >= 50_000 * 256  # two k-halves
    assert lib.mi_oov_topk_catalogue_prepare(None, 50_000, 129, None, None) == -2
    assert lib.mi_oov_topk_catalogue_prepare(None, 50_000, 64, None, None) == -1
    assert lib.mi_oov_score_topk_prepared(None, 4, None, 50_000, 64, 20, 0, None, None, None, None, None, None, None) == -1
    pw = lib.mi_oov_score_topk_prepared_workspace(4096, 10_000_000, 64, 2, 0)                 # lists only: the catalogue holds the bf16 copy of E
    assert 0 < pw < (2 << 30) < lib.mi_oov_score_topk_workspace(4096, 10_000_000, 2) and pw < lib.mi_oov_score_topk_prepared_workspace(4096, 10_000_000, 64, 2, 1)
    assert lib.mi_oov_score_topk_workspace_d(4096, 10_000_000, 64, 2) < (2 << 30) < lib.mi_oov_score_topk_workspace_d(4096, 10_000_000, 128, 2) == lib.mi_oov_score_topk_workspace(4096, 10_000_000, 2)  # one / two k-halves of E's copy: the host's 2 GiB chunking threshold lies between
    assert lib.mi_oov_score_topk_prepared_workspace(4096, 1000, 64, 20, 0) == 0 and lib.mi_oov_score_topk_prepared_workspace(4096, 50_000, 129, 20, 0) == 0
    assert lib.mi_oov_score_topk_prepared(None, 4, None, 1000, 64, 20, 0, None, None, None, None, None, None, None) == -2
    assert lib.mi_oov_linear_act(None, 4, 16, None, None, 8, 7, None, None) in (-2, -3)       # unknown activation
    assert lib.mi_oov_linear_x3_weights_bytes(512, 1024) == 64 * 512 * 96                     # [K/16][N -> 256s][3 planes x 32 B]
    assert lib.mi_oov_linear_x3_weights_bytes(33, 70) == 5 * 256 * 96 and lib.mi_oov_linear_x3_weights_bytes(0, 16) == -2
    assert lib.mi_oov_linear_x3_prepare(None, 512, 1024, None, None) == -1 and lib.mi_oov_linear_x3_prepare(None, 512, 0, None, None) == -2
    assert lib.mi_oov_linear_x3(None, 4, 16, None, None, 8, 7, None, None) == -3              # unknown activation
    assert lib.mi_oov_linear_x3(None, 4, 16, None, None, 8, 1, None, None) == -1 and lib.mi_oov_linear_x3(None, 0, 16, None, None, 8, 1, None, None) == 0
    assert lib.mi_oov_last_hip_error() == 0                                                  # nothing touched the GPU


def test_argument_validation_of_the_round4_entry_points(lib):
    """Evaluation rows / duplicates / metric sums, the dense exclusion route, the fused backward and the fused bucketing: sizes
    and error codes, all checked before any HIP call."""
    assert lib.mi_oov_eval_rows_build(None, -1, None, None, None, 5, None, None, None, None, None) == -2
    assert lib.mi_oov_eval_rows_build(None, 3, None, None, None, 5, None, None, None, None, None) == -1   # seg_ptr NULL
    assert lib.mi_oov_segment_dedup(None, None, 0, None, None) == 0 and lib.mi_oov_segment_dedup(None, None, 3, None, None) == -1
    assert lib.mi_oov_segment_dedup(None, None, -1, None, None) == -2
    assert lib.mi_oov_topk_hits_range(None, 3, 0, None, None, 0, 10, None, None) == -2
    assert lib.mi_oov_topk_hits_range(None, 3, 10, None, None, 0, 10, None, None) == -1
    assert lib.mi_oov_topk_metric_sums_workspace(36000, 10) >= 6 * 36000 * 10 * 8 + 36000 and lib.mi_oov_topk_metric_sums_workspace(0, 10) == 0
    assert lib.mi_oov_topk_metric_sums(None, 5, 300, None, None, None, 0, 1, None, None, None, None) == -2      # K > 256
    assert lib.mi_oov_topk_metric_sums(None, 5, 10, None, None, None, 0, 2, None, None, None, None) == -2       # sides: 1 or 3
    assert lib.mi_oov_topk_metric_sums(None, 5, 10, None, None, None, 0, 1, None, None, None, None) == -1
    assert lib.mi_oov_score_topk_excl_dense_workspace(100, 1000) >= 100 * 1000 * 4 + 100 * 16 * 8
    assert lib.mi_oov_score_topk_excl_dense_workspace(0, 1000) == 0
    assert lib.mi_oov_score_topk_excl_dense(None, 4, None, 0, 64, 10, 0, None, None, None, None, None, None) == -2
    assert lib.mi_oov_score_topk_excl_dense(None, 4, None, 100, 64, 10, 0, None, None, None, None, None, None) == -1
    assert lib.mi_oov_score_topk_excl_dense(None, 0, None, 100, 64, 10, 0, None, None, None, None, None, None) == 0
    assert lib.mi_oov_lsh_backward_fused_workspace(65536, 9, 64) == 1024 * 16 * 64             # planes padded to groups of eight
    assert lib.mi_oov_lsh_backward_fused_counters() >= 17
    assert lib.mi_oov_lsh_embed_backward_fused(None, None, 5, 8, 300, None, None, None, None) == -2
    assert lib.mi_oov_lsh_embed_backward_fused(None, None, 5, 8, 64, None, None, None, None) == -1
    assert lib.mi_oov_slsh_embed_backward_fused(None, None, 5, 0, 64, None, None, None, None) == -2
    assert lib.mi_oov_bucket_by_owner_scratch() >= 17
    assert lib.mi_oov_bucket_by_owner_fused(None, 5, 100, 50, 17, 8, -1, None, None, None, None, None, None, None) == -2   # world > 16
    assert lib.mi_oov_bucket_by_owner_fused(None, 0, 100, 50, 2, 8, -1, None, None, None, None, None, None, None) == -2    # empty batch
    assert lib.mi_oov_bucket_by_owner_fused(None, 5, 100, 50, 2, 8, 2, None, None, None, None, None, None, None) == -2     # my_rank >= world
    assert lib.mi_oov_bucket_by_owner_fused(None, 5, 100, 50, 2, 8, -1, None, None, None, None, None, None, None) == -1
    assert lib.mi_oov_last_hip_error() == 0


def test_header_is_plain_c(tmp_path):
    """include/mi_oov.h must be consumable from C (the drop-in boundary has no C++ or torch types)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_header.c"
    src.write_text('#include "mi_oov.h"\nint main(void) { return mi_oov_version() == 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", f"-I{root}/include", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
