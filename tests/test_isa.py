"""Static checks on the gfx950 ISA of the hand-scheduled kernels (tools/isa_check.py): hipcc cross-compiles without a GPU.

* csrc/gemm8.hip: every wait of its LDS-DMA pipeline is a COUNTED vmcnt, which is only meaningful while nothing but
  LDS-DMA is in the vector-memory queue.  A compiler-inserted register spill or hoisted global access inside the
  pipelined region would silently break that; the checker walks the control-flow graph of every instantiation.
* csrc/tp_comm.hip: remote pulls of the tensor-parallel exchange are single 16-byte system-scope loads.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import isa_check  # noqa: E402


def test_gemm8_counted_waits_see_only_lds_dma():
    """Also: no static LDS, LDS-DMA in the scalar-base form, (STORE / RESID / SwiGLU builds) nothing but 16-byte stores in the
    epilogue, fed by v_permlane16_swap — and the literal immediate of every counted wait in a steady-state loop equals that
    loop's own LDS-DMA count per K-tile (`four` of the schedule model in tests/test_host_logic.py::test_gemm8_schedule): an
    edited wait_vm<> count in csrc/gemm8.hip fails HERE (checked by mutation: FOUR - 1 in one phase -> 10 errors)."""
    report, errors = isa_check.check_gemm8()
    assert len(report) == 16, report                      # 4 epilogues x 4 tile configurations
    assert not errors, "\n".join(errors)
    for name, n_dma, n_wait, _ in report:
        assert n_dma >= 20 and n_wait >= 12, (name, n_dma, n_wait)


def test_attention_matrix_blocks_are_software_pipelined():
    """Also: no spills, no static LDS, no packed fp32 VALU, no canonicalising maxima, and no compiler-counted vmcnt wait inside a
    tile loop (the LDS-DMA requests of the next tile share that queue)."""
    report, errors = isa_check.check_attention()
    assert not errors, "\n".join(errors)
    assert report and report[0][2] >= 6 * 32   # S and P·V blocks of the six (groups, order) instantiations


def test_tp_pull_transport_uses_16_byte_system_scope_loads():
    report, errors = isa_check.check_tp_pull()
    assert not errors, "\n".join(errors)
    assert {r[0].split("ILi")[-1][:1] for r in report if "reduce" in r[0]} >= {"2", "4", "8"}
