"""CPU tests of the host-side logic: C-ABI surface, parameter packing, the prior DataLoader
protocol, schedules / samplers against the reference's recorded values, the train() loop plumbing
(with the CPU oracle standing in for the HIP model), and the data-parallel helpers over gloo."""
import ctypes
import os
import random
import re
import subprocess
import sys

import pytest
import torch
from torch import nn

from oracle import pfn_oracle
from transformerscandobayesianinference_amd import _hip, dp, utils
from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')



def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return str(sk.getsockname()[1])

def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'pfn_hip.h')).read()
    declared = set(re.findall(r'\b(pfn_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib = _hip.lib()           # loads without a GPU; resolves and type-checks every symbol
    for name in declared:
        assert hasattr(lib, name)
    assert lib.pfn_abi_version() == _hip.ABI_VERSION


def test_c_client_of_the_abi(tmp_path):
    """The boundary is a C ABI: include/pfn_hip.h compiles as C99, and a C program (tests/cabi_check.c) dlopens the library, resolves every declared entry point
    and runs the host-side ones -- no Python, no torch types anywhere in the signatures."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    header = open(os.path.join(ROOT, 'include', 'pfn_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(pfn_[a-z0-9_]+)\s*\(', header)))
    exe = str(tmp_path / 'cabi_check')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'cabi_check.c'), '-ldl', '-o', exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    res = subprocess.run([exe, _hip.LIB_PATH] + declared, capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'cabi_check ok' in res.stdout


def test_attention_kernels_keep_their_register_and_lds_budgets(tmp_path):
    """The attention kernels are written against hard budgets: no scratch in any operand-precision variant, at most 256 VGPRs where two
    waves share a SIMD (8-wave configurations: head dims 32 / 64 / 128) and at most 512 where one wave has it (head dim 256); the LDS budget (160 KiB) is a
    static_assert in the launchers, so it holds whenever this compiles.  Read off
    the compiler's own kernel descriptors (hipcc -S of attention.hip with the flags of csrc/build.sh) -- a change that starts spilling fails here, not in a profile."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'transformerscandobayesianinference_amd', 'csrc', 'attention.hip')
    asm = str(tmp_path / 'attention.s')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-S', '--cuda-device-only', src, '-o', asm], check=True,
                   capture_output=True)
    text = open(asm).read()
    seen = 0
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
        name, body = m.group(1), m.group(2)
        if not ('DF16b' in name or 'DF16_' in name) or 'attn_' not in name:          # the 16-bit (operand precision) instantiations: bf16 and fp16
            continue
        vgpr = int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1))
        scratch = int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1))
        assert scratch == 0, (name, scratch)
        head_dim = re.search(r'Li(\d+)E', name)
        if head_dim:                                              # (the delta kernel has no head-dim parameter)
            assert vgpr <= (512 if int(head_dim.group(1)) == 256 else 256), (name, vgpr)
        seen += 1
    assert seen >= 2 * (4 * 2 * 3 + 1)    # forward / key-block pass / query-block pass x 4 head dims x {plain, dropout}, + delta; x {bf16, fp16}


def test_step_gemm_kernels_have_no_scratch(tmp_path):
    """The same for the GEMMs the training step launches (bench.py's kernel table names them): the 256 x 256 NT kernel in its default tile mode <flags, 2, 64>, the
    LayerNorm-fused forms at 8 waves (emsize 512) and their wide variants, the grouped weight-gradient kernel -- no scratch, at most 256 VGPRs (two waves per SIMD).
    (Alternative tile modes are tuning options outside the step and are not held to it.)"""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'transformerscandobayesianinference_amd', 'csrc', 'gemm.hip')
    asm = str(tmp_path / 'gemm.s')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', asm], check=True, capture_output=True)
    step = re.compile(r'gemm_nt_big_kernelIDF16[b_]Li\d+ELi2ELi64E|gemm_nt_ln_kernelIDF16[b_]Li8E|gemm_nt_lnbwd_kernelIDF16[b_]Li8E|gemm_nt_ln_wide_kernel|gemm_nt_lnbwd_wide_kernel|gemm_tn_big_kernel')
    seen = 0
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', open(asm).read(), re.S):
        name, body = m.group(1), m.group(2)
        if not step.search(name):
            continue
        assert int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)) == 0, name
        assert int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)) <= 256, name
        seen += 1
    assert seen >= 2 * (10 + 4 + 2 + 3 + 1)      # x {bf16, fp16}


def test_param_layout_matches_reference_state_dict_order():
    lib = _hip.lib()
    d = _hip.ModelDesc(18, 512, 4, 1024, 6, 1000, _hip.PREC_BF16, 1e-5)
    n = lib.pfn_param_layout(ctypes.byref(d), None, None, 0)
    assert n == 4 + 12 * 6 + 4
    offs, nums = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
    assert lib.pfn_param_layout(ctypes.byref(d), offs, nums, n) == n
    assert sum(nums) == 14_177_768      # parameters of the north-star model (14.18 M)
    assert all(o % 64 == 0 for o in offs) and all(offs[i] + nums[i] <= offs[i + 1] for i in range(n - 1))
    assert list(nums[:4]) == [512 * 18, 512, 512, 512] and list(nums[4:8]) == [3 * 512 * 512, 3 * 512, 512 * 512, 512]
    assert lib.pfn_param_count(ctypes.byref(d)) >= offs[n - 1] + nums[n - 1]
    assert lib.pfn_workspace_bytes(ctypes.byref(d), 8, 2000) > 0
    bad = _hip.ModelDesc(18, 200, 2, 200, 6, 100, _hip.PREC_BF16, 1e-5)   # head dim 100 (reference train() default)
    assert lib.pfn_param_count(ctypes.byref(bad)) < 0 and b'head dim' in lib.pfn_last_error_string()


def test_top_layer_row_rule_and_flop_accounting():
    """The stack drops the top encoder layer's train rows (the reference returns output[single_eval_pos:], transformer.py:91) under a rule that is
    host logic -- pfn_top_layer_rows -- and bench.py's FLOP count follows it."""
    import bench
    lib = _hip.lib()
    d = _hip.ModelDesc(18, 512, 4, 1024, 6, 1000, _hip.PREC_BF16, 1e-5)
    B, S = 32, 2000
    rows = lambda sep, drop=0, desc=d: lib.pfn_top_layer_rows(ctypes.byref(desc), B, S, sep, drop)
    assert rows(1604) == (S - 1604) * B and rows(500) == (S - 500) * B          # test rows only
    assert rows(499) == B * S and rows(0) == B * S and rows(S) == B * S          # short train part / no test row: every row
    # the schedule travels in the descriptor (ABI 6): a forward and its backward cannot disagree about the row layout
    all_rows = _hip.ModelDesc(18, 512, 4, 1024, 6, 1000, _hip.PREC_BF16, 1e-5, 0.0, _hip.SCHED_TOP_LAYER_ALL_ROWS)
    assert rows(1604, 0, all_rows) == B * S
    ws = lambda desc: lib.pfn_workspace_bytes(ctypes.byref(desc), B, S)
    assert ws(all_rows) < ws(d)                                                  # ... and the compact-row buffers are only carved when they can be used
    try:      # the test / profiling knob changes the DEFAULT handed to new descriptors, not calls on existing ones
        assert lib.pfn_default_schedule() == 0
        assert lib.pfn_set_tuning(6, 0) == 0 and lib.pfn_default_schedule() == _hip.SCHED_TOP_LAYER_ALL_ROWS and rows(1604) == (S - 1604) * B
    finally:
        lib.pfn_set_tuning(6, 1)
    assert lib.pfn_default_schedule() == 0
    bad_bits = _hip.ModelDesc(18, 512, 4, 1024, 6, 1000, _hip.PREC_BF16, 1e-5, 0.0, 1 << 20)
    assert rows(1604, 0, bad_bits) < 0
    with_dropout = _hip.ModelDesc(18, 512, 4, 1024, 6, 1000, _hip.PREC_BF16, 1e-5, 0.2)
    assert rows(1604, 1, with_dropout) == B * S and rows(1604, 0, with_dropout) == B * S   # dropout keeps the full-layout row indices (the compact-row buffers are not carved)
    assert ws(with_dropout) > 0
    # profiling hook: nothing recorded unless enabled; reading an empty slot is fine without a GPU
    import ctypes as _c
    ms, n = _c.c_double(-1), _c.c_int64(-1)
    assert lib.pfn_profile_read(4, _c.byref(ms), _c.byref(n)) == 0 and ms.value == 0.0 and n.value == 0
    assert lib.pfn_profile_read(999, _c.byref(ms), _c.byref(n)) < 0
    no_layers = _hip.ModelDesc(18, 512, 4, 1024, 0, 1000, _hip.PREC_BF16, 1e-5)
    assert rows(1604, 0, no_layers) == B * S
    assert rows(S + 1) < 0
    # FLOPs: the reference's graph against what the result needs -- the difference is the top layer's train rows behind the K / V projection
    nf, E, F, L, O, sep = 18, 512, 1024, 6, 1000, 1604
    full, need = bench.fwd_flops(S, sep, nf, E, F, L, O), bench.fwd_flops(S, sep, nf, E, F, L, O, True)
    saved = 2 * sep * E * E + 4 * E * sep * sep + 2 * sep * E * E + 4 * sep * E * F      # Q projection, attention, out_proj, FFN of the sep train rows
    assert full - need == saved and 0.85 < need / full < 0.92
    assert bench.fwd_flops(S, sep, nf, E, F, 0, O, True) == bench.fwd_flops(S, sep, nf, E, F, 0, O)


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under the package imports, opens or executes anything under oracle/ or /root/reference, and bench.py
    reaches the oracle only inside parity_check / cpu_baseline / parity_inputs / oracle_loss_and_means (after the timed window)."""
    import ast
    pkg = os.path.join(ROOT, 'transformerscandobayesianinference_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                text = open(os.path.join(dirpath, f)).read()
                assert '/root/reference' not in text, f
                if f.endswith('.py'):
                    for node in ast.walk(ast.parse(text)):
                        names = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ''] if isinstance(node, ast.ImportFrom) else []
                        assert not any(n == 'oracle' or n.startswith('oracle.') for n in names), (f, names)
    tree = ast.parse(open(os.path.join(ROOT, 'bench.py')).read())
    allowed = {'parity_check', 'cpu_baseline', 'parity_inputs', 'oracle_loss_and_means'}
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or '').split('.')[0] == 'oracle' for n in ast.walk(fn))
        assert not uses or fn.name in allowed, fn.name
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and 'oracle' in ast.dump(n) for n in tree.body)      # no module-level import either


def test_bench_profile_classes_match_the_header():
    """bench.py reads the library's in-step kernel timings by slot number (pfn_profile_read): its table must be the header's PFN_PROF_* enum, and every
    kernel class the step launches must have a slot (include/pfn_hip.h)."""
    import bench
    header = open(os.path.join(ROOT, 'include', 'pfn_hip.h')).read()
    enum = {k: int(v) for k, v in re.findall(r'PFN_PROF_([A-Z0-9_]+) = (\d+)', header)}
    slots = enum.pop('SLOTS')
    want = {'ATTN_FWD': 'attn_fwd', 'ATTN_BWD_DELTA': 'attn_bwd_delta', 'ATTN_BWD_KV': 'attn_bwd_kv', 'ATTN_BWD_DQ': 'attn_bwd_dq', 'GEMM_QKV': 'gemm_qkv',
            'GEMM_OUT_LN': 'gemm_out_proj_ln', 'GEMM_LIN1': 'gemm_linear1_gelu', 'GEMM_LIN2_LN': 'gemm_linear2_ln', 'GEMM_DHPRE': 'gemm_dhpre', 'GEMM_DY1': 'gemm_dy1_lnbwd',
            'GEMM_DCTX': 'gemm_dctx', 'GEMM_DX': 'gemm_dx_lnbwd', 'WGRAD': 'gemm_tn_group'}
    assert set(enum) == set(want)
    assert {want[k]: v for k, v in enum.items()} == bench.PROF_SLOTS
    assert max(enum.values()) + 2 == slots and all(v % 2 == 0 for v in enum.values())     # slot + 1 = the top layer's launch on the test rows


def test_one_image_layout_of_the_key_block_pass_is_bank_conflict_free():
    """attention.hip BwdKvCfg::ONE: query q of a 32-query tile sits in LDS row perm(q) (the two 2-bit fields of the row index swapped) under the row
    padding RB + 16 bytes.  Under the LDS model of MI355X_MICROARCH.md (64 banks of 4 bytes; ds_read_b128 served in the 16-lane groups listed there,
    ds_read_b64_tr_b16 in two 32-lane groups; lanes of one group must hit distinct banks) both access patterns of the pass are conflict-free and
    read the bytes the MFMA operands want."""
    perm = lambda q: 4 * (q & 3) + ((q >> 2) & 3) + (q & ~15)
    assert sorted(perm(q) for q in range(32)) == list(range(32)) and all(perm(perm(q)) == q for q in range(32))
    groups128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups128 += [[l + 32 for l in g] for g in groups128]
    for D in (32, 64, 128, 256):
        stride = D * 2 + 16                                   # bytes (bf16 rows, PadStride::ROW)
        # row reads (S = Q K^T, dP = dO V^T): lane (h, li) reads 16 bytes of query li at k-step kk: LDS row perm(li), chunk 2 kk + h
        for kk in range(D // 16):
            for g in groups128:
                banks = []
                for lane in g:
                    h, li = lane >> 5, lane & 31
                    a = perm(li) * stride + (2 * kk + h) * 16
                    banks += [((a // 4) + d) % 64 for d in range(4)]
                assert len(set(banks)) == 64, (D, kk)
        # transposed reads (dV^T += dO^T P, dK^T += Q^T dS): lane (h, g, i) reads 8 bytes of rows 4 (i >> 2) + h + k0 and that + 2
        for k0 in (0, 16):
            for col0 in range(0, D, 32):
                for h in (0, 1):
                    for second in (0, 2):
                        banks, queries = [], set()
                        for g in (0, 1):
                            for i in range(16):
                                row = 4 * (i >> 2) + h + k0 + second
                                a = row * stride + (col0 + 16 * g + 4 * (i & 3)) * 2
                                banks += [(a // 4) % 64, (a // 4 + 1) % 64]
                                queries.add(perm(row))
                        assert len(set(banks)) == 64, (D, k0, col0)
                        # the four rows are the queries the accumulator layout pairs with this half-wave: k0 + 4 h + {0..3} (+ 8 for the second read)
                        assert queries == {k0 + 4 * h + j + (8 if second else 0) for j in range(4)}


def test_model_state_dict_keys_match_reference():
    from transformerscandobayesianinference_amd import bar_distribution, encoders
    from transformerscandobayesianinference_amd.transformer import TransformerModel
    rec = torch.load(os.path.join(GOLD, 'model_small_h32.pt'))
    cfg = rec['config']
    m = TransformerModel(encoders.Linear(cfg['F'], cfg['E']), cfg['nbars'], cfg['E'], cfg['H'], cfg['nhid'], cfg['L'], 0.0,
                         y_encoder=encoders.Linear(1, cfg['E']))
    m.criterion = bar_distribution.FullSupportBarDistribution(rec['state_dict']['criterion.borders'])
    assert list(m.state_dict().keys()) == list(rec['state_dict'].keys())
    m.load_state_dict(rec['state_dict'])
    for layer in TransformerModel(encoders.Linear(2, 64), 4, 64, 2, 64, 2, y_encoder=encoders.Linear(1, 64)).transformer_encoder.layers:
        assert layer.linear2.weight.abs().sum() == 0 and layer.self_attn.out_proj.weight.abs().sum() == 0   # transformer.py:49-53
    with pytest.raises(_hip.HipExtensionError):   # CPU tensors never fall back to PyTorch
        m((rec['x'], rec['y']), single_eval_pos=5)
    # the model descriptor handed to the C ABI (include/pfn_hip.h pfn_model_desc): dropout is part of it, inference defaults to f32 kernels
    d = TransformerModel(encoders.Linear(3, 64), 8, 64, 2, 128, 2, 0.25, y_encoder=encoders.Linear(1, 64))
    desc = d._make_desc()
    assert (desc.num_features, desc.emsize, desc.nhead, desc.nhid, desc.nlayers, desc.n_out) == (3, 64, 2, 128, 2, 8)
    assert abs(desc.dropout - 0.25) < 1e-7 and desc.precision == _hip.PREC_FP16 and d.eval_precision == 'f32'
    assert d._make_desc('f32').precision == _hip.PREC_F32
    assert _hip.lib().pfn_workspace_bytes(desc, 2, 64) > _hip.lib().pfn_workspace_bytes(m._make_desc(), 2, 64) * 0   # (host-only size query works without a GPU)
    s1, s2 = d._next_dropout_seed(), d._next_dropout_seed()
    assert s1 != s2 and 0 <= s1 < 2 ** 64


def test_schedules_and_samplers_match_reference_values():
    rec = torch.load(os.path.join(GOLD, 'utils.pt'))
    p = nn.Parameter(torch.zeros(1))
    for name, fn in [('cosine', utils.get_cosine_schedule_with_warmup), ('linear', utils.get_linear_schedule_with_warmup)]:
        opt = torch.optim.SGD([p], lr=1.0)
        sch = fn(opt, 5, 20)
        vals = []
        for _ in range(22):
            vals.append(sch.get_last_lr()[0])
            opt.step()
            sch.step()
        assert vals == pytest.approx(rec[name], abs=1e-12)
        assert vals[0] == 0.0     # lr is 0 during the first epoch
    random.seed(123)
    s = utils.get_weighted_single_eval_pos_sampler(50)
    assert [s() for _ in range(200)] == rec['weighted_draws']
    random.seed(123)
    s = utils.get_uniform_single_eval_pos_sampler(50)
    assert [s() for _ in range(200)] == rec['uniform_draws']
    assert utils.get_openai_lr(nn.Linear(10, 10)) == pytest.approx(rec['openai_lr_110'])
    from transformerscandobayesianinference_amd.transformer import TransformerModel
    assert torch.equal(TransformerModel.generate_D_q_matrix(6, 2), rec['d_q_mask_6_2'])


def test_bucket_limits_and_cli_dict_action():
    from transformerscandobayesianinference_amd.bar_distribution import get_bucket_limits
    rec = torch.load(os.path.join(GOLD, 'bar_distribution.pt'))
    assert torch.allclose(get_bucket_limits(8, full_range=(-2., 6.)), rec['bucket_limits_uniform'])
    ys = torch.randn(1003, generator=torch.Generator().manual_seed(1))
    lim = get_bucket_limits(10, ys=ys)
    assert lim[0] == ys[:1000].min() and lim[-1] == ys[:1000].max()
    counts = torch.histc(ys[:1000], bins=10, min=-10, max=10)  # noqa: F841  (smoke)
    inner = torch.bucketize(ys[:1000], lim[1:-1])
    assert torch.bincount(inner, minlength=10).tolist() == [100] * 10      # equal-count buckets
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--kw', action=utils.StoreDictKeyPair, nargs='+')
    ns = ap.parse_args(['--kw', 'num_features=5', 'name=gp', 'hps=(1e-4,1.0,0.6)', 'evil=__import__("os").system("true")'])
    assert ns.kw == {'num_features': 5, 'name': 'gp', 'hps': (1e-4, 1.0, 0.6), 'evil': '__import__("os").system("true")'}


def test_prior_dataloader_protocol():
    calls = []

    def get_batch(batch_size, seq_len, num_features, hyperparameters=None):
        calls.append((batch_size, seq_len, num_features))
        return pfn_oracle.get_batch_fast_gp(batch_size, seq_len, num_features, hyperparameters)

    DL = get_batch_to_dataloader(get_batch)
    DL.num_outputs = 1
    dl = DL(num_steps=3, batch_size=4, seq_len=9, num_features=2, hyperparameters=(0.1, 0.1, 0.1))
    assert len(dl) == 3 and dl.num_features == 2 and dl.num_outputs == 1 and dl.fuse_x_y is False
    items = list(dl)
    assert len(items) == 3 and calls == [(4, 9, 2)] * 3
    (x, y), t = items[0]
    assert x.shape == (9, 4, 2) and y.shape == (9, 4) and t.shape == (9, 4)
    fused = DL(num_steps=1, fuse_x_y=True, batch_size=2, seq_len=5, num_features=3)
    xf, tf = next(iter(fused))
    assert xf.shape == (5, 2, 4) and torch.equal(xf[0, :, -1], torch.zeros(2)) and torch.equal(xf[1:, :, -1], tf[:-1])
    assert isinstance(dl, torch.utils.data.DataLoader)
    assert DL.get_batch_method(2, 5, 3)[0].shape == (5, 2, 3)      # the notebook calls it unbound


class _OracleModel(nn.Module):
    """Stand-in for TransformerModel in the CPU test of the train() plumbing: same constructor and
    forward signature, arithmetic by the oracle (tests only; the product model has no CPU path)."""

    def __init__(self, encoder, n_out, ninp, nhead, nhid, nlayers, dropout=0.0, y_encoder=None, pos_encoder=None, decoder=None,
                 input_normalization=False, precision='bf16', deterministic=False):
        super().__init__()
        from transformerscandobayesianinference_amd.transformer import _EncoderParams
        self.encoder, self.y_encoder, self.nhead = encoder, y_encoder, nhead
        self.transformer_encoder = _EncoderParams(ninp, nhid, nlayers)
        self.decoder = nn.Sequential(nn.Linear(ninp, nhid), nn.GELU(), nn.Linear(nhid, n_out))
        self._flat = self._grad = None
        self.nlayers, self._custom_decoder, self._first_group_hook = nlayers, False, None      # what dp.OverlappedGradientReducer asks of a model

    def _fused_embedding(self):
        return True

    def layer_offset(self, l):
        """first element of encoder layer l in the flat buffers (state-dict order: embedding | layers | decoder), as TransformerModel.layer_offset"""
        self.flat_parameters()
        o = 0
        for name, p in self.named_parameters():
            if name.startswith(f'transformer_encoder.layers.{l}.') or (l >= self.nlayers and name.startswith('decoder.')):
                return o
            o += p.numel()
        return o

    def flat_parameters(self):
        if self._flat is None:
            ps = list(self.parameters())
            self._flat = torch.cat([p.detach().reshape(-1) for p in ps])
            self._grad = torch.zeros_like(self._flat)
            o = 0
            for p in ps:
                p.data = self._flat[o:o + p.numel()].view(p.shape)
                p.grad = self._grad[o:o + p.numel()].view(p.shape)
                o += p.numel()
        return self._flat, self._grad

    def mark_params_updated(self):
        pass

    def forward(self, src, src_mask=None, single_eval_pos=None):
        sd = {k: v for k, v in self.named_parameters()}
        return pfn_oracle.forward(sd, src[0], src[1], single_eval_pos, self.nhead, dtype=torch.float32)


class _TorchAdam(torch.optim.Adam):
    def __init__(self, model, lr, max_grad_norm=1.0):
        super().__init__(model.parameters(), lr=lr)
        self.model, self.grad_multiplier = model, 1.0

    def step(self, zero_grad=False):
        for p in self.model.parameters():
            p.grad.mul_(self.grad_multiplier)
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
        super().step()
        if zero_grad:
            for p in self.model.parameters():
                p.grad.zero_()

    def skipped_steps(self):      # (FusedClipAdam's counter of steps with a non-finite gradient: train() reads it once per epoch)
        return 0


def _cpu_train(monkeypatch, **kw):
    from transformerscandobayesianinference_amd import bar_distribution, encoders, train as train_mod
    monkeypatch.setattr(train_mod, 'TransformerModel', _OracleModel)
    monkeypatch.setattr(train_mod, 'FusedClipAdam', _TorchAdam)

    class CpuBar(bar_distribution.FullSupportBarDistribution):
        def forward(self, logits, y):
            return pfn_oracle.bar_nll(logits, y, self.borders, True)

    DL = get_batch_to_dataloader(lambda batch_size, seq_len, num_features, hyperparameters=None:
                                 pfn_oracle.get_batch_fast_gp(batch_size, seq_len, num_features, hyperparameters))
    DL.num_outputs = 1
    borders = bar_distribution.get_bucket_limits(10, ys=pfn_oracle.get_batch_fast_gp(50, 10, 2)[1])
    return train_mod.train(DL, CpuBar(borders), encoders.Linear, emsize=32, nhid=32, nlayers=1, nhead=1, dropout=0.0,
                           y_encoder_generator=encoders.Linear, extra_prior_kwargs_dict={'num_features': 2, 'fuse_x_y': False},
                           single_eval_pos_gen=utils.get_weighted_single_eval_pos_sampler(10), bptt=12, verbose=False, **kw)


def test_train_loop_plumbing_on_cpu(monkeypatch):
    torch.manual_seed(0)
    random.seed(0)
    loss, pos, model = _cpu_train(monkeypatch, epochs=3, steps_per_epoch=4, batch_size=4, lr=1e-2, warmup_epochs=1, aggregate_k_gradients=2)
    assert isinstance(loss, float) and loss == loss and len(pos) == 12
    assert next(model.parameters()).device.type == 'cpu'
    with pytest.raises(AssertionError):
        _cpu_train(monkeypatch, epochs=1, steps_per_epoch=3, batch_size=4, aggregate_k_gradients=2)


def test_train_loop_replays_the_reference_train_golden_on_cpu(monkeypatch):
    """The training LOOP pinned to the reference's own `train.train` (train.py:58-110,134; utils.py:10-22):
    tests/golden/train_loop_small.pt holds recorded batches, a recorded eval-position stream and what the reference's loop made of them --
    4 epochs x 8 batches, aggregate_k_gradients = 2 (micro-batch gradients SUMMED per optimizer step), cosine schedule with one
    warm-up epoch stepped per EPOCH whose first epoch runs at lr = 0 (Q5).  This repo's train() replays the same stream here with the f32
    oracle standing in for the HIP model (the GPU suite runs the real stack on the same fixture): every batch loss, every learning rate
    and the final weights must come out as the reference's did."""
    import replay
    from transformerscandobayesianinference_amd import bar_distribution, encoders, train as train_mod
    rec = torch.load(os.path.join(GOLD, 'train_loop_small.pt'))
    monkeypatch.setattr(train_mod, 'TransformerModel', _OracleModel)
    monkeypatch.setattr(train_mod, 'FusedClipAdam', _TorchAdam)

    class CpuBar(bar_distribution.FullSupportBarDistribution):
        def forward(self, logits, y):
            return pfn_oracle.bar_nll(logits, y, self.borders, True)

    losses, lrs, total, final = replay.replay(train_mod.train, rec, CpuBar, encoders, utils.get_cosine_schedule_with_warmup)
    cfg = rec['config']
    assert len(losses) == cfg['epochs'] * cfg['steps_per_epoch']
    assert lrs == pytest.approx(rec['batch_lr'], rel=1e-12, abs=0)
    assert lrs[0] == 0.0 and lrs[cfg['steps_per_epoch']] == cfg['lr']                   # epoch 1 trains at lr = 0 (reference quirk Q5)
    assert max(abs(a - b) / abs(b) for a, b in zip(losses, rec['batch_losses'])) < 2e-5
    assert abs(total - rec['returned_total_loss']) < 2e-5 * abs(rec['returned_total_loss'])
    assert replay.update_error(final, rec) < 1e-3


_DP_SCRIPT = r'''
import os, sys, random, torch
sys.path.insert(0, sys.argv[1])
from transformerscandobayesianinference_amd import dp
rank, world, local = dp.init_from_env(backend='gloo')
W = int(os.environ['WORLD_SIZE'])
assert world == W and dp.world_size() == W and dp.rank() == rank and dp.local_batch_size(8 * W) == 8
seed = dp.seed_ranks()
shared = [random.random() for _ in range(3)]           # python stream (single_eval_pos) is rank-shared
own = torch.rand(3)                                    # torch stream (prior draws) is rank-distinct
gathered = [None] * W
torch.distributed.all_gather_object(gathered, (shared, own.tolist()))
assert all(g[0] == gathered[0][0] for g in gathered) and len({tuple(g[1]) for g in gathered}) == W
# gradient all-reduce of the flat buffer: mean over ranks == gradient of the global batch
torch.manual_seed(0)
w = torch.randn(5, requires_grad=True)
data = torch.arange(8 * W, dtype=torch.float32).view(4 * W, 2)[rank * 4:(rank + 1) * 4]
loss = ((data @ w[:2]) ** 2).mean()
loss.backward()
flat = w.grad.clone()
dp.all_reduce_gradients(flat)
flat /= world
full = torch.arange(8 * W, dtype=torch.float32).view(4 * W, 2)
w2 = w.detach().clone().requires_grad_(True)
((full @ w2[:2]) ** 2).mean().backward()
assert torch.allclose(flat, w2.grad, rtol=1e-6), (flat, w2.grad)
# the two-collective reducer of the training loop (tail = upper layers + decoder first, then the head) on a host buffer
buf = torch.arange(10, dtype=torch.float32) * (rank + 1)
red = dp.OverlappedGradientReducer(flat_grad=buf, split=6, first_group_layers=1)
red.arm(1)
assert red.armed()
red.finish()
assert torch.equal(buf, torch.arange(10, dtype=torch.float32) * (W * (W + 1) // 2)) and not red.overlapped_last_step and not red.armed()
assert red.layout() == dict(total_bytes=40, overlapped_bytes=16, exposed_bytes=24, first_group_layers=1)
whole = dp.OverlappedGradientReducer(flat_grad=torch.ones(4) * (rank + 1))      # no split: one collective
whole.finish()
assert torch.equal(whole.grad, torch.ones(4) * (W * (W + 1) // 2))

# ---- the training loop itself on W ranks (CPU control path: the f64 oracle stands in for the HIP model, as in test_train_loop_plumbing_on_cpu) ----
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import test_host as th
from oracle import pfn_oracle
from transformerscandobayesianinference_amd import bar_distribution, encoders, utils, train as train_mod
from transformerscandobayesianinference_amd.priors.utils import get_batch_to_dataloader
train_mod.TransformerModel, train_mod.FusedClipAdam = th._OracleModel, th._TorchAdam

class CpuBar(bar_distribution.FullSupportBarDistribution):
    def forward(self, logits, y):
        return pfn_oracle.bar_nll(logits, y, self.borders, True)

seen = []      # (batch size this rank was asked for, first feature value of the draw)
def get_batch(batch_size, seq_len, num_features, hyperparameters=None):
    x, y, t = pfn_oracle.get_batch_fast_gp(batch_size, seq_len, num_features, hyperparameters)
    seen.append((batch_size, float(x[0, 0, 0])))
    return x, y, t
DL = get_batch_to_dataloader(get_batch)
DL.num_outputs = 1
torch.manual_seed(5)
borders = bar_distribution.get_bucket_limits(10, ys=pfn_oracle.get_batch_fast_gp(50, 10, 2)[1])      # (same on every rank: drawn before the ranks are seeded apart)
dp.seed_ranks(1234)
torch.manual_seed(99)      # ... but identical INITIAL weights come from the broadcast in train(), not from this seed: make the seeds differ on purpose below
torch.manual_seed(99 + rank)
random.seed(7)             # rank-shared eval positions
loss, pos, model = train_mod.train(DL, CpuBar(borders), encoders.Linear, emsize=32, nhid=32, nlayers=2, nhead=1, dropout=0.0, y_encoder_generator=encoders.Linear,
                                   extra_prior_kwargs_dict={'num_features': 2, 'fuse_x_y': False}, single_eval_pos_gen=utils.get_weighted_single_eval_pos_sampler(10),
                                   bptt=12, verbose=False, epochs=2, steps_per_epoch=4, batch_size=2 * W, lr=1e-2, warmup_epochs=1, aggregate_k_gradients=2)
assert all(b == 2 for b, _ in seen) and len(seen) == 8          # every rank draws batch_size / world datasets per step
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
everyone = [None] * W
torch.distributed.all_gather_object(everyone, (flat.tolist(), [v for _, v in seen], loss))
assert all(e[0] == everyone[0][0] for e in everyone), 'the ranks ended with different weights'      # same initial weights (broadcast) + same averaged gradients
assert len({tuple(e[1]) for e in everyone}) == W, 'two ranks drew the same data'
assert all(abs(e[2] - everyone[0][2]) < 1e-12 for e in everyone)                                     # the returned loss is the mean over ranks
# one step by hand: the reducer's two collectives over W ranks give the gradient of the GLOBAL batch
m = th._OracleModel(encoders.Linear(2, 32), 10, 32, 1, 32, 2, y_encoder=encoders.Linear(1, 32))
fl, gr = m.flat_parameters()
torch.distributed.broadcast(fl, 0)
crit = CpuBar(borders)
x, y, t = pfn_oracle.get_batch_fast_gp(3, 12, 2)      # this rank's shard (rank-distinct torch stream)
red = dp.OverlappedGradientReducer(m)
assert red.first_group_layers == 1 and red.split == m.layer_offset(1) and 0 < red.split < gr.numel()
red.arm(1)
crit(m((x, y), single_eval_pos=8).reshape(-1, 10), t[8:].flatten()).mean().backward()
red.finish()
shards = [None] * W
torch.distributed.all_gather_object(shards, (x, y, t))
m2 = th._OracleModel(encoders.Linear(2, 32), 10, 32, 1, 32, 2, y_encoder=encoders.Linear(1, 32))
fl2, gr2 = m2.flat_parameters()
fl2.copy_(fl)
X, Y, T = (torch.cat([s[i] for s in shards], 1) for i in range(3))
crit(m2((X, Y), single_eval_pos=8).reshape(-1, 10), T[8:].flatten()).mean().backward()
assert torch.allclose(gr / W, gr2, rtol=1e-4, atol=1e-7), (gr / W - gr2).abs().max()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world', [2, 4, 8])
def test_data_parallel_helpers_and_train_loop_over_gloo(tmp_path, world):
    """SURVEY.md 8(e) on the CPU control path at world sizes 2, 4 and 8: rank seeding (shared eval-position stream, distinct draws),
    local_batch_size, the two-collective reducer, and train() itself -- every rank draws batch_size / world datasets per step, the ranks end with identical
    weights, and one hand-run step's reduced gradient equals the gradient of the concatenated global batch."""
    script = tmp_path / 'dp_check.py'
    script.write_text(_DP_SCRIPT)
    _PORT = _free_port()      # (a fixed port collided once in a full-suite run: a listener of an earlier test was still closing)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_PORT, OMP_NUM_THREADS='1')
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
                          '--master-port', _PORT, str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count('ok') == world


def test_gp_mix_hyperprior_moments():
    """Gamma hyper-priors of priors.fast_gp_mix (reference :26,33,37): E[lengthscale] = 3/6, E[outputscale] = .5/.15,
    noise ~ Gamma(1.1, .05) floored at 1e-4."""
    from transformerscandobayesianinference_amd.priors import fast_gp_mix
    g = torch.Generator().manual_seed(0)
    ls, osc, nz = fast_gp_mix.sample_hyperparameters(40000, 2, None, 'cpu', generator=g)
    assert ls.shape == (40000, 2) and abs(ls.mean().item() - 0.5) < 0.01 and abs(ls.var().item() - 3 / 36) < 0.005
    assert abs(osc.mean().item() - 0.5 / 0.15) < 0.1
    assert abs(nz.mean().item() - 22.0) < 0.5 and nz.min().item() >= 1e-4
    ls2, _, nz2 = fast_gp_mix.sample_hyperparameters(1000, 3, {'lengthscale_concentration': 2.0, 'lengthscale_rate': 4.0, 'noise_concentration': 0.01, 'noise_rate': 100.0}, 'cpu', generator=g)
    assert abs(ls2.mean().item() - 0.5) < 0.05 and (nz2 >= 1e-4).all() and (nz2 == 1e-4).any()


def _tabular_config(prior_type='mlp', causal=False):
    from transformerscandobayesianinference_amd.priors.utils import scaled_beta_sampler_f
    cfg = {'prior_type': prior_type, 'prior_is_causal': causal,
           'prior_nlayers_sampler': {'const 3': (lambda: 3)}, 'prior_emsize_sampler': {'beta': scaled_beta_sampler_f(2., 4., 30, 2)},
           'prior_activations': torch.nn.Tanh, 'prior_sigma_gamma_k': 3.6, 'prior_sigma_gamma_theta': 0.07,
           'prior_noise_std_gamma_k': 1.9, 'prior_noise_std_gamma_theta': 0.05, 'prior_dropout_sampler': {'none': (lambda: 0.0)},
           'prior_num_features_used_sampler': {'beta': scaled_beta_sampler_f(1., 1.6, 12, 2)}, 'prior_order_y': True,
           'prior_normalize_by_used_features': False,
           'prior_lengthscale_concentration': 1.2, 'prior_nu': 2.5, 'prior_outputscale_concentration': 0.8, 'prior_y_minmax_norm': False,
           'prior_noise_concentration': 1.1, 'prior_noise_rate': 400.0, 'prior_noise': 0.1, 'prior_outputscale': 0.7, 'prior_lengthscale': 0.4,
           'prior_outputscale_mean': 0.7, 'prior_outputscale_std_f': 0.1, 'prior_lengthscale_mean': 0.4, 'prior_lengthscale_std_f': 0.1,
           'emsize': 64, 'nhead': 2, 'nhid_factor': 2, 'nlayers': 2, 'dropout': 0.0, 'batch_size': 8, 'bptt': 40, 'lr': 1e-3, 'epochs': 1, 'num_features': 12}
    return cfg


def test_tabular_hyperparameter_builders():
    """tabular.get_*_prior_hyperparameters (reference tabular.py:47-106): tuple / dict layouts the priors unpack."""
    from transformerscandobayesianinference_amd import tabular
    hps = tabular.get_mlp_prior_hyperparameters(_tabular_config())
    assert len(hps) == 17 and hps[0]() == 3 and hps[2] is torch.nn.Tanh and hps[6] is True and hps[9] is False
    assert hps[8] is None and hps[10] is None and hps[15] is None and hps[16] == 0.0 and hps[13] is True and hps[14] is False
    assert 2 <= hps[1]() <= 32 and hps[3]() > 0 and hps[4]() > 0 and hps[5]() == 0.0 and 2 <= hps[7]() <= 14
    mix = tabular.get_gp_mix_prior_hyperparameters(_tabular_config('gp_mix'))
    assert mix['nu'] == 2.5 and mix['noise_rate'] == 400.0
    assert mix['y_minmax_norm'] == 1.2 and mix['categorical_data'] is False          # the reference's crossed keys, kept
    gp = tabular.get_gp_prior_hyperparameters(_tabular_config('gp'))
    assert len(gp) == 7 and gp[0] == 0.1 and gp[1]() == 0.7 and gp[2]() == 0.4 and gp[3] is True
    meta = tabular.get_meta_gp_prior_hyperparameters(_tabular_config('custom_gp_mix'))
    assert len(meta) == 7 and 0.3 < meta[1]() < 1.1 and 0.2 < meta[2]() < 0.6
    cls, h, extra = tabular._prior_for(_tabular_config('gp'))
    assert cls.__name__ == 'DL' or hasattr(cls, 'get_batch_method')
    assert h == (0.1, 0.7, 0.4) and extra == {}
    with pytest.raises(ValueError):
        tabular._prior_for({'prior_type': 'nope'})
    assert tabular.get_uniform_single_eval_pos_sampler(5)() in range(5)


def test_ridge_prior_and_baseline_vs_sklearn():
    """priors.ridge: draw shapes / statistics, and `evaluate` against the per-dataset sklearn fits
    the reference loops over (priors/ridge.py:22-34)."""
    from sklearn.linear_model import Ridge
    from transformerscandobayesianinference_amd.priors import ridge
    torch.manual_seed(0)
    x, y, clean = ridge.get_batch(6, 24, 3, noisy_std=.01, device='cpu')
    assert x.shape == (24, 6, 3) and y.shape == clean.shape == (24, 6) and 0 <= x.min() and x.max() < 1
    assert (y - clean).std().item() < 0.02 and clean.abs().max().item() < 1.5
    for alpha in (1e-3, 0.5):
        got, secs = ridge.evaluate(x, y, clean, alpha=alpha)
        want = [0.]
        for t in range(1, 24):
            sq = 0.
            for b in range(6):
                fit = Ridge(alpha=alpha).fit(x[:t, b].numpy(), y[:t, b].numpy())
                sq += (fit.predict(x[t, b].unsqueeze(0).numpy())[0] - clean[t, b].item()) ** 2
            want.append(sq / 6)
        assert got.shape == (24,) and torch.allclose(got.double(), torch.tensor(want, dtype=torch.float64), rtol=1e-4, atol=1e-7)
    dl = ridge.DataLoader(num_steps=2, batch_size=4, seq_len=10, num_features=3, device='cpu')
    (xx, yy), tt = next(iter(dl))
    assert xx.shape == (10, 4, 3) and yy.shape == tt.shape == (10, 4) and dl.num_outputs == 1


def test_bar_distribution_eval_methods_match_reference_golden():
    """quantile / mode / ei (reference bar_distribution.py:40-80) against values recorded from the
    reference classes themselves (tests/golden/bar_distribution.pt, oracle/make_golden.py::bar_case).  These are host
    tensor arithmetic in both code bases (vectorised here, Python loops there): bit-exact."""
    import os
    from transformerscandobayesianinference_amd import bar_distribution as bd
    rec = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bar_distribution.pt'))
    seen = 0
    for key, c in rec.items():
        if not isinstance(key, tuple):
            continue
        nb, full = key
        crit = (bd.FullSupportBarDistribution if full else bd.BarDistribution)(c['borders'].clone())
        lg = c['logits']
        assert torch.equal(crit.quantile(lg), c['quantile']) and torch.equal(crit.quantile(lg, center_prob=.9), c['quantile90'])
        assert torch.equal(crit.mode(lg), c['mode'])
        assert torch.equal(crit.ei(lg, c['best_f'], maximize=True), c['ei_max'])
        assert torch.equal(crit.ei(lg, c['best_f'], maximize=False), c['ei_min'])
        assert torch.equal(crit.quantile(lg.view(8, 8, nb)), c['quantile'].view(8, 8, 2))      # leading dims are kept
        seen += 1
    assert seen == 5
    assert torch.equal(bd.get_bucket_limits(8, full_range=(-2., 6.)), rec['bucket_limits_uniform'])


def test_train_places_the_prior_on_the_training_device():
    """train() hands its device to priors that take one (under data parallelism every rank owns one GPU and the
    priors default to cuda:0) and leaves custom get_batch functions without a device argument alone."""
    from transformerscandobayesianinference_amd.train import _accepts_kwarg
    from transformerscandobayesianinference_amd.priors import binarized_regression, fast_gp, fast_gp_mix, mlp, ridge
    for cls in (fast_gp.DataLoader, fast_gp_mix.DataLoader, mlp.DataLoader, ridge.DataLoader, binarized_regression.Binarized_fast_gp_dataloader):
        assert _accepts_kwarg(cls.get_batch_method, 'device')
    assert not _accepts_kwarg(lambda batch_size, seq_len, num_features: None, 'device')
    assert not _accepts_kwarg(None, 'device')


def test_checkpoint_tuple_roundtrip_on_cpu(tmp_path):
    """The notebooks' `(state_dict, optimizer_state)` checkpoint tuple (tabular.save_checkpoint / load_checkpoint) and the
    Student-t interval helper of the evaluation sweeps: host-only plumbing."""
    from transformerscandobayesianinference_amd import evaluation, tabular
    torch.manual_seed(0)
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    path = str(tmp_path / 'ck.cpkt')
    tabular.save_checkpoint(a, path, optimizer_state={'step': 7})
    state, opt = torch.load(path)
    assert set(state) == {'weight', 'bias'} and opt == {'step': 7} and not state['weight'].requires_grad
    assert tabular.load_checkpoint(b, path) == {'step': 7}
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
    torch.save((a.state_dict(), None), path)                    # what the reference notebooks write
    assert tabular.load_checkpoint(b, path) is None
    mean, half = evaluation.compute_mean_and_conf_interval([1., 2., 3., 4.])
    assert mean == 2.5 and abs(half - 2.0541) < 1e-3


def test_ragged_batch_layout():
    """Host side of forward_batches / pfn_stack_forward_ragged: per-dataset eval positions and the dataset-major compact test-row offsets."""
    from transformerscandobayesianinference_amd.transformer import ragged_layout
    seps, per_dataset, offs = ragged_layout(10, [2, 1, 3], [7, -2, 0])
    assert seps == [7, 8, 0]                                   # -2 counts from the end, as single_eval_pos does through slicing in the reference
    assert per_dataset == [7, 7, 8, 0, 0, 0]
    assert offs == [0, 3, 6, 8, 18, 28, 38] and offs[-1] == sum(10 - s for s in per_dataset)
    seps, per_dataset, offs = ragged_layout(10, [1, 1], [10, 25])      # no test rows at all (clamped)
    assert seps == [10, 10] and offs == [0, 0, 0]


def test_split_precision_products_keep_f32_accuracy_in_the_blocked_cholesky():
    """The numerical basis of the GP sampler's fp16 products (gp_prior.hip: gp_syrk_planes_kernel, gp_trsm_wide_kernel), on the CPU: the blocked factorisation is
    emulated in f32 with the rank-256 update -- and the wide solve's block products -- computed from two fp16 terms per operand on a power-of-two scale (three
    products).  On the ill-conditioned north-star matrices (5 features, noise 1e-4) the draw y = L z stays as close to the f64 factorisation as with exact f32
    products; two bf16 terms (the cheaper split one might reach for) are several times worse or fail outright.  (tools/sim_gp_split.py: the tool that priced the
    kernels before they were written.)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import sim_gp_split as sim
    torch.manual_seed(1)
    for T in (768, 1280):
        e = sim.errors(T, 5, 1e-4, 1.0, 0.6, 'rbf', modes=('f32', 'fp16x3', 'fp16x3+solve-fp16x3', 'bf16x3'))
        assert e['f32'] < 2e-3, e
        assert e['fp16x3'] < 1.5 * e['f32'] and e['fp16x3+solve-fp16x3'] < 1.5 * e['f32'], e
        assert not (e['bf16x3'] < 2 * e['f32']), e            # nan (failed factorisation) or clearly worse
