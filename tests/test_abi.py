"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/echoscene_hip.h declares, and the host-side packers produce the documented layouts.
No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    import __graft_entry__ as ge
    ge.build()
    from echoscene_amd import hip
    return hip.lib()


def test_every_declared_symbol_is_exported(L):
    from echoscene_amd import hip
    hdr = open(os.path.join(ROOT, 'include', 'echoscene_hip.h')).read()
    declared = set(re.findall(r'\b(es_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 20
    raw = C.CDLL(hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), 'libechoscene_hip.so lacks %s' % name
    assert declared == set(hip.EXPORTS), 'ctypes table and header disagree: %s' % (declared ^ set(hip.EXPORTS))
    assert L.es_abi_version() == 3


def test_struct_sizes_match_header(L, tmp_path):
    """sizeof() of every ABI struct as seen by the C compiler == the ctypes mirror."""
    import subprocess
    from echoscene_amd import hip
    names = {'es_seg': hip.Seg, 'es_linear_args': hip.LinearArgs, 'es_update_args': hip.UpdateArgs,
             'es_conv_args': hip.ConvArgs, 'es_gn_args': hip.GNArgs, 'es_ln_args': hip.LNArgs,
             'es_attn_args': hip.AttnArgs, 'es_geglu_args': hip.GegluArgs, 'es_copy_args': hip.CopyArgs,
             'es_tocl_args': hip.ToClArgs, 'es_stem_args': hip.StemArgs, 'es_vq_args': hip.VQArgs,
             'es_rowsel_args': hip.RowSelArgs,
             'es_op': hip.Op}
    src = '#include <stdio.h>\n#include "echoscene_hip.h"\nint main(){' + ''.join(
        'printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + 'return 0;}'
    c = tmp_path / 'sz.c'
    c.write_text(src)
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(c), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n, cls in names.items():
        assert sizes[n] == C.sizeof(cls), '%s: C %d vs ctypes %d' % (n, sizes[n], C.sizeof(cls))


def test_pack_linear_layout(L):
    rs = np.random.RandomState(0)
    N, K = 21, 38
    W = rs.standard_normal((N, K)).astype(np.float32)
    n = L.es_pack_linear_f32_size(N, K)
    assert n == 2 * 3 * 256
    out = np.zeros(n, np.float32)
    assert L.es_pack_linear_f32(W.ctypes.data, N, K, out.ctypes.data) == 0
    o = out.reshape(2, 3, 64, 4)
    for nt in range(2):
        for kb in range(3):
            for lane in range(64):
                j, q = lane & 15, lane >> 4
                for e in range(4):
                    nn, kk = nt * 16 + j, kb * 16 + 4 * q + e
                    exp = W[nn, kk] if (nn < N and kk < K) else 0.0
                    assert o[nt, kb, lane, e] == exp


def test_holders_refuse_to_compute():
    from echoscene_amd.model.graph import GraphTripleConvNet
    net = GraphTripleConvNet(8, 4, num_layers=1, hidden_dim=8, residual=True, mlp_normalization='batch')
    with pytest.raises(RuntimeError, match='parameter container'):
        net(torch.zeros(1, 8))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under echoscene_amd/ or model/ may reference it."""
    bad = []
    for base in ('echoscene_amd', 'model'):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith('.py'):
                    txt = open(os.path.join(d, f)).read()
                    if re.search(r'^\s*(from|import)\s+oracle\b', txt, re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad
