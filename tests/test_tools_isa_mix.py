"""tools/isa_mix.py: instruction classes and kernel extraction on a hand-written assembly snippet (no hipcc, no GPU)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_mix", os.path.join(ROOT, "tools", "isa_mix.py"))
isa_mix = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_mix)

ASM = """
\t.text
_ZN1a4kernEv:                           ; @_ZN1a4kernEv
; %bb.0:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tv_mov_b32_e32 v1, 0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tds_read_b128 v[4:7], v2
\tv_mfma_f32_16x16x32_f16 v[8:11], v[4:7], v[4:7], v[8:11]
\tv_exp_f32_e32 v3, v3
\tv_cvt_pk_f16_f32 v3, v3, v3
\tv_fma_f32 v3, v3, v3, v3
\tv_accvgpr_read_b32 v4, a0
\ts_cbranch_scc1 .LBB0_1
\tbuffer_load_dwordx4 v2, s[0:3], 0 offen lds
\ts_endpgm
\t.amdhsa_next_free_vgpr 12
; ScratchSize: 0
; Occupancy: 8
_ZN1a5otherEv:
\ts_endpgm
"""


def test_classify():
    c = isa_mix.classify
    assert c("v_mfma_f32_32x32x16_f16") == "mfma"
    assert c("v_exp_f32_e32") == "valu.transcendental" and c("v_rcp_f32_e32") == "valu.transcendental"
    assert c("v_cvt_pk_f16_f32") == "valu.cvt" and c("v_max3_f32") == "valu.minmax"
    assert c("v_fma_f32") == "valu.fma32" and c("v_pk_fma_f32") == "valu.packed"
    assert c("v_accvgpr_write_b32") == "valu.mov" and c("v_mul_lo_u32") == "valu.int"
    assert c("ds_read_b128") == "lds" and c("buffer_load_dwordx4") == "vmem"
    assert c("s_waitcnt") == "s.wait" and c("s_cbranch_scc1") == "s.branch" and c("s_mul_i32") == "salu"


def test_kernel_body_and_meta():
    name, body, meta = isa_mix.kernel_body(ASM, "4kern")
    assert name == "_ZN1a4kernEv"
    assert body[-1].strip() == "s_endpgm" and not any("5other" in l for l in body)
    assert meta["vgprs"] == 12 and meta["scratch"] == 0
    ops = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
    assert sum(isa_mix.classify(o) == "mfma" for o in ops) == 1
    assert sum(isa_mix.classify(o).startswith("valu.") for o in ops) == 5
