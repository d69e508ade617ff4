"""Build-level checks that need no GPU: hipcc's gfx950 output of the fused decoder kernels is scanned for a miscompilation pattern
(register-allocator copies ahead of an exec-mask restore, tools/check_exec_restore.py) and for spills."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def decoder_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = {}
    d = tmp_path_factory.mktemp("asm")
    for name in ("decoder", "decoder_bwd"):
        dst = str(d / (name + ".s"))
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S",
                            "--cuda-device-only", os.path.join(ROOT, "uni3detr_amd", "csrc", name + ".hip"), "-o", dst,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = (dst, r.stderr)
    return out


def test_no_register_copies_ahead_of_exec_restore(decoder_asm):
    import check_exec_restore as chk
    for name, (path, _) in decoder_asm.items():
        hits = chk.scan(path)
        assert not hits, (name, hits[:5])


def test_checker_recognises_the_pattern(tmp_path):
    import check_exec_restore as chk
    p = tmp_path / "bad.s"
    p.write_text("k:  ; @k\n.LBB0_1:\n\tv_add_f32_e32 v1, v2, v3\n\ts_andn2_b64 exec, exec, s[4:5]\n\ts_cbranch_execnz .LBB0_1\n.LBB0_2:\n"
                 "\tv_accvgpr_write_b32 a5, v7\n\ts_or_b64 exec, exec, s[6:7]\n\ts_endpgm\n")
    assert [h[2] for h in chk.scan(str(p))] == ["v_accvgpr_write_b32 a5, v7"]
    q = tmp_path / "ok.s"
    q.write_text("k:  ; @k\n.LBB0_2:\n\ts_or_b64 exec, exec, s[6:7]\n\tv_accvgpr_write_b32 a5, v7\n\ts_endpgm\n")
    assert chk.scan(str(q)) == []


def test_fused_decoder_kernels_do_not_spill(decoder_asm):
    for name, (_, remarks) in decoder_asm.items():
        for kind in ("ScratchSize \\[bytes/lane\\]", "VGPRs Spill"):
            vals = [int(v) for v in re.findall(kind + r": (\d+)", remarks)]
            assert vals and max(vals) == 0, (name, kind, vals)
