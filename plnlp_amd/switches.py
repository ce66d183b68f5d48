"""The switches of the Python host, in ONE table -- INTEGRATION.md section 5 is generated from it
(`python -m plnlp_amd.switches` prints the section; tests/test_host_logic.py checks that the document, this table and
the `os.environ` reads in the code agree).  The C library reads no environment.

ENV: read once, at import, by the module named.  MODULE: a module-level dict the tests and A/B scripts flip in process
(`ops.STEP_THROTTLE["depth"] = 0`); these have NO environment variable -- setting `PLNLP_<NAME>` does nothing."""

ENV = [
    # (variable, read by, default, effect)
    ("PLNLP_GEMM_MATH", "ops.GEMM_MATH['mode']", "bf16x3",
     "`f32`: dense products on the f32-input MFMA (an fmaf chain, the reference's sgemm arithmetic) instead of three-term "
     "bf16 splits"),
    ("PLNLP_GEMM_STATIONARY_B", "ops.GEMM_STATIONARY_B['enabled']", "1",
     "`0`: every split-bf16 product on the 128x128 tile kernel (no stationary pre-split weights, `csrc/gemm_x3s.hip`); "
     "same bits"),
    ("PLNLP_GEMM_BLOCK", "ops.GEMM_BLOCK['mode']", "auto",
     "the stationary-weights product with a whole 256-row block of the result per workgroup (`csrc/gemm_x3b.hip`, one persistent "
     "workgroup per CU): `auto` = launches of >= 32 768 rows at 224-column tiles (a layer 193..224 wide: citation2's h = 200), "
     "`off` = never, `all` = the 256-column tiles too (measured equal to `gemm_x3s` within -2..+5 %: both run at the power limit), "
     "`nolead` / `all-nolead` = without the leading half blocks of odd workgroups; same bits"),
    ("PLNLP_EDGE_SEGMENT", "ops.EDGE_SEGMENT['form']", "auto",
     "the fused scorer's deterministic backward over the touched nodes (`edge_segment_bwd_group_kernel`): `auto` = long segments are "
     "shared by the waves of a workgroup -- from 8 192 segments on, eight segments per workgroup of eight waves, those of more than 64 "
     "items summed by all eight (citation2: 263 K nodes with two items each and a few hubs with hundreds), below that every segment by "
     "the four waves of its own workgroup (ddi: 4 267 nodes, 123 items on average); `wave` = one wave per segment whatever its length "
     "(the round-5 form: the launch lasts as long as the hub's chain); `group4` = groups of four waves / four segments; `noslab` = few segments without the XCD-pinned column slabs (ddi: eight "
     "workgroups per segment, one eighth of the columns each, so that every XCD's L2 holds one slab of the table)"),
    ("PLNLP_GEMM_WIDE_WGRAD", "ops.GEMM_WIDE_WGRAD['enabled']", "1",
     "`0`: every weight gradient on the 128x128 tile kernel instead of the whole-block kernel (`csrc/gemm_wgw.hip`: one 224-wide "
     "block for a 129..224 wide layer -- citation2's h = 200 --, 256x256 blocks for collab's 256x512 pair and ddi's 512x512; "
     "reductions of >= 32 768 rows); same arithmetic, another summation order"),
    ("PLNLP_EMB_PAD", "ops.EMB_PAD['floats']", "16",
     "row granule (floats) an embedding table of an unaligned width is padded to when it is kept padded "
     "(`model.PAD_EMBEDDING_TABLE`): `4` = 16-byte rows (citation2: 52 columns), `16` = whole 64-byte sectors (64 columns)"),
    ("PLNLP_DENSE_AGG", "ops.DENSE_AGG['enabled']", "1",
     "`0`: the aggregation of a dense graph (>= 5 % of all node pairs, 2 048..16 384 nodes: ogbl-ddi) on the CSR kernels like every "
     "other graph, instead of as a bf16-counts x split-bf16 product on the matrix cores (`csrc/aggregate_dense.hip`); another "
     "summation order"),
    ("PLNLP_SPARSE_FORWARD", "ops.SPARSE_FORWARD['enabled']", "1",
     "`0`: the last conv of a training step is evaluated at every node instead of the rows the batch touches"),
    ("PLNLP_AGG_AUTOTUNE", "ops.AGG_AUTOTUNE['enabled']", "1",
     "`0`: the aggregation always runs one wave per row with position chunks for hub rows (no choice among the slab "
     "forms / the source-range hub form)"),
    ("PLNLP_AGG_FORM", "ops.AGG_AUTOTUNE['force']", "unset",
     "`<int>`: pins the aggregation form (the `PLNLP_AGG_*` flag bits of `include/plnlp_hip.h`) instead of choosing it "
     "(counter-collection runs)"),
    ("PLNLP_AGG_FORMS_FILE", "ops.AGG_FORMS['path']", "plnlp_amd/agg_forms.json",
     "where the chosen aggregation form per (graph shape, width) is kept: a shape found there is NOT re-measured, so two "
     "boxes sum in the same order and give the same bits; `none`: measure per process (round-4 behaviour)"),
    ("PLNLP_CAPTURE", "capture.CAPTURE['enabled']", "0",
     "`1`: `BaseModel.train` / `bench.py` replay the step from two hipGraphs (`plnlp_amd/capture.py`): host work per step "
     "1.06 -> 0.25 ms, GPU time +2 %, bit-identical results; `available`: only flips ROCm's graph-packet-capture flag at "
     "import (the test session)"),
    ("PLNLP_HIP_LIB", "_lib.LIB_PATH", "plnlp_amd/libplnlp_hip.so",
     "path of another build of the same ABI (same-box A/B runs of a kernel variant)"),
    ("PLNLP_BENCH_DEADLINE_S", "bench.py launcher", "1500",
     "`bench.py --gpus N`: overall deadline after which the remaining ranks are stopped"),
]

MODULE = [
    # (module attribute, default, effect of the other setting)
    ("model.FUSE_EMBEDDING_ADAM['enabled']", "True",
     "`False`: the embedding's gradient is materialised and stepped by the optimiser kernel (`emb.weight.grad` is then "
     "set); same bits"),
    ("model.STREAM_PERMUTATION['enabled']", "True",
     "`False`: `BaseModel.train` draws the whole epoch permutation with `torch.randperm` before the first step (same "
     "permutation)"),
    ("model.FUSE_LOSS_ACC['enabled']", "True",
     "`False`: the epoch's loss sum as three element-wise launches per step, as `model.py:169` does it"),
    ("model.SHARD_SPARSE['enabled']", "True",
     "`False`: the sharded step (`dp_exchange='shard'`) runs its last layer over the whole row block"),
    ("model.PAD_EMBEDDING_TABLE['enabled']", "True",
     "`False`: an embedding table of an unaligned width under a first GCN layer (citation2: 50) is a plain contiguous tensor, "
     "padded for the aggregation and un-padded for Adam by a copy each step, instead of living padded (`PLNLP_EMB_PAD`)"),
    ("ops.STEP_THROTTLE['depth']", "2",
     "steps the host may run ahead of the GPU; `0`: unbounded, side-stream tensors handed over with `record_stream`"),
    ("ops.PROLOGUE_OVERLAP['enabled']", "True",
     "`False`: a batch's edge pre-processing runs in line on the main stream"),
    ("ops.SPARSE_BACKWARD['enabled'] / ['max_expected_fraction']", "True / 0.97",
     "row-sparse backward of the last conv and the touched fraction of the nodes up to which it is used"),
    ("ops.SPLIT_THRESHOLD", "0",
     "hub-row chunk length; `0` = 128 up to 2^20 source rows, 1024 beyond"),
    ("ops.AGG_FUSED['enabled']", "True",
     "`False`: the long rows' chunk pass as its own launch"),
    ("ops.GCN_INPUT_FUSION['enabled']", "True",
     "`False`: the first GCN layer transforms `[emb ‖ x]` first, as GCNConv does, instead of aggregating first"),
    ("ops.FUSE_EDGE_MLP['enabled']", "False",
     "`True`: MLPPredictor's Hadamard formed inside the GEMM loaders (measured slower)"),
    ("ops.FUSE_HEAD_FORWARD['enabled']", "True",
     "`False`: MLPPredictor's 1-output head as its own pass over the hidden activation (`plnlp_matvec_f32`) instead of in the "
     "hidden product's epilogue (`PLNLP_EPI_ROWDOT`)"),
    ("ops.FUSE_HEAD_BACKWARD['enabled']", "True",
     "`False`: the backward of MLPPredictor's 1-output head as four passes over the hidden activation (outer product, three "
     "column sums) instead of one (`plnlp_mlp_head_backward_f32`)"),
    ("ops.COLSUM_IN_WGRAD['enabled']", "True",
     "`False`: a conv's bias gradient always as its own pass over dz (`plnlp_colsum_f32`) instead of out of the whole-block "
     "weight-gradient kernel, which stages every row of dz anyway (`plnlp_gemm_operand.a_colsum`); another summation order"),
    ("ops.COLSUM_SIDE_STREAM['enabled']", "True",
     "`False`: a conv's bias gradient (column sums of dz) in line on the main stream instead of on the second side stream "
     "beside the weight-gradient GEMM (-1.0 % on the collab step)"),
    ("ops.SIDE_STREAM_PRIORITY['value']", "-1",
     "HIP priority of the side streams (`0` normal, `1` low: both measured +0.5 %)"),
    ("ops.SPLIT_K_SLOTS['slots']", "512",
     "workgroup slots one round of a split-K launch fills (768 measured +-5 % depending on the shape)"),
    ("ops.EDGE_BACKWARD['mode']", "segment",
     "anything else: the endpoint-gather backward by atomics instead of the deterministic segmented reduction"),
]

FOREIGN = [
    ("DEBUG_CLR_GRAPH_PACKET_CAPTURE (ROCm's own)", "set to `0` by `import plnlp_amd` when unset, HIP not yet initialised "
     "and `PLNLP_CAPTURE` is `1` / `available`",
     "with ROCm 7.0's graph packet capture ON a replayed hipGraph faults after any device-to-host copy "
     "(profiles/r03_capture_debug.md); when the runtime was initialised with it on, the step pipeline stays eager "
     "(`plnlp_amd.GRAPH_REPLAY_SAFE`)"),
]

BEGIN, END = "<!-- switches:begin (generated: python -m plnlp_amd.switches) -->", "<!-- switches:end -->"


def markdown() -> str:
    out = [BEGIN, "", "Environment variables (read once, at import):", "",
           "| variable | lands in | default | effect |", "|---|---|---|---|"]
    out += ["| `%s` | `%s` | `%s` | %s |" % row for row in ENV]
    out += ["| `%s` | %s | %s |" % ((FOREIGN[0][0],) + (FOREIGN[0][1] + " | —", FOREIGN[0][2]))]
    out += ["", "In-process switches (module dicts; **no environment variable** — flip them from Python, as the tests and the "
                "A/B scripts do):", "", "| attribute (`plnlp_amd.`) | default | effect |", "|---|---|---|"]
    out += ["| `%s` | `%s` | %s |" % row for row in MODULE]
    out += ["", END]
    return "\n".join(out)


if __name__ == "__main__":
    print(markdown())
