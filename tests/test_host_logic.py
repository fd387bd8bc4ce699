"""Host-side logic (no GPU): schedules, sequence layout, prompt templates — checked against the reference's own
formulas (generators/parallel_generator.py:73-99,157-159,318-321; inference.py:117-161; SURVEY.md §8d numbers)."""
import math
import os

import pytest
import torch

from mmada_parallel_amd import synth
from mmada_parallel_amd.generators import interleave_generator as ig
from mmada_parallel_amd.generators import parallel_generator as pg
from mmada_parallel_amd.utils import (SPECIAL_TOKENS, add_break_line, build_ti2ti_sequence, calculate_vq_params,
                                      generate_text_image_to_text_image_prompt)


def test_num_transfer_tokens_a_variant():
    for total, steps in [(256, 128), (256, 100), (16, 8), (16, 7), (5, 9), (0, 4)]:
        m = torch.zeros(1, 300, dtype=torch.bool)
        m[0, :total] = True
        got = pg.get_num_transfer_tokens(m, steps)[0].tolist()
        remaining, want = total, []
        for s in range(steps):  # reference :88-97
            t = max(0, remaining - int(total * (1 - (s + 1) / steps)))
            want.append(t)
            remaining -= t
        assert got == want and sum(got) == total


def test_num_transfer_tokens_m_variant():
    m = torch.zeros(2, 64, dtype=torch.bool)
    m[0, :50] = True
    m[1, :7] = True
    got = ig.get_num_transfer_tokens(m, 8)
    assert got[0].tolist() == [7, 7, 6, 6, 6, 6, 6, 6] and got[1].tolist() == [1] * 7 + [0]


def test_image_step_schedule_and_mask_len():
    assert len(set(pg.image_step_indices(128, 64))) == 64 and min(pg.image_step_indices(128, 64)) == 32
    assert len(set(pg.image_step_indices(32, 16))) == 16
    ml = pg.mask_len_schedule(1024, 128)
    assert ml[-1] == -1  # fp32 cos(pi/2) < 0 -> exactly one token is left for the random fill
    for s in (0, 17, 63, 126):
        r = (s + 1) / 128
        assert ml[s] == int(math.floor(1024 * float(torch.cos(torch.tensor(r) * math.pi / 2))))


def test_sequence_layout_matches_survey_numbers():
    job = synth.synthetic_job(512, 512, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    ids = job["input_ids"][0].tolist()
    assert len(ids) == 2438 and job["image_start"] == 1124 and job["text_start"] == 2181 and job["text_end"] == 2437
    assert job["seq_len"] == 1024 and job["newline_every"] == 32
    job1 = synth.synthetic_job(256, 256, text_gen_length=256, prompt_len=64, uncond_prompt_len=24, seed=1)
    assert job1["input_ids"].shape[1] == 1654 and job1["seq_len"] == 256 and job1["newline_every"] == 16


def test_build_ti2ti_sequence():
    T = SPECIAL_TOKENS
    prompt, unc = list(range(100, 110)), list(range(200, 205))
    seq_len, nl_every, gh, gw = calculate_vq_params(64, 64)
    img = [T["boi"]] + add_break_line([T["image_token_offset"] + i for i in range(seq_len)], gh, gw, T["newline_token"]) + [T["eoi"]]
    s = build_ti2ti_sequence(prompt, unc, img, 64, 64, text_gen_length=12, end_token_ids=[T["answer_end"]])
    ids = s["input_ids"]
    assert ids[:9] == prompt[:-1] and ids[9:9 + len(img)] == img and ids[9 + len(img)] == prompt[-1]
    assert ids[s["code_start"]] == T["answer_start"] and ids[s["image_start"] - 1] == T["boi"]
    assert ids[s["image_end"]] == T["eoi"] and s["text_start"] == s["image_end"] + 1
    span = ids[s["image_start"]:s["image_end"]]
    assert span.count(T["mask_token"]) == seq_len and span.count(T["newline_token"]) == gh
    assert ids[s["text_start"]:s["text_end"]] == [T["mask_token"]] * 12 and ids[-1] == T["answer_end"]
    assert s["uncon_text"] == unc[:-1] + img + unc[-1:] and s["uncon_image"] == prompt
    assert len(s["uncon_text"]) < s["code_start"]  # the uncond prefix overwrite stays inside the conditioning part


def test_prompt_template():
    a, b = generate_text_image_to_text_image_prompt("make it red", "SYS")
    assert a == "<system>SYS</system><user>make it red</user>" and b == "<system>SYS</system><user><uncondition></user>"


def test_vq_state_dict_keys_match_the_reference_modules():
    """Checkpoint contract of the MAGVITv2 path: the key names and shapes the loader expects (synth.vq_*_param_shapes,
    mirrored by the slot table of csrc/vq_decoder.hip) are those of the reference's VQGANDecoder / VQGANEncoder built with
    their default arguments (recorded by oracle/gen_golden.py from the reference modules' own state_dict())."""
    import os

    import numpy as np

    from conftest import GOLDEN
    from mmada_parallel_amd import synth

    for fixture, shapes in (("vq_decode.npz", synth.vq_decoder_param_shapes(synth.VQ_CFG_M)),
                            ("vq_encode.npz", synth.vq_encoder_param_shapes(synth.VQ_ENC_CFG_M))):
        ref = [str(x) for x in np.load(os.path.join(GOLDEN, fixture))["full_keys"]]
        mine = [f"{k}:{tuple(v)}" for k, v in shapes.items()]
        assert sorted(ref) == sorted(mine)


def test_generate_image_keep_schedule_matches_the_oracle_rule():
    import torch

    from mmada_parallel_amd.generators.image_generation_generator import keep_schedule
    from oracle import generate_image_oracle as gio

    for n, steps in ((1024, 18), (256, 7), (16, 5), (3, 4)):
        want = []
        for step in range(steps):
            if step < steps - 1:
                frac = gio.cosine_schedule(torch.tensor([(step + 1) / steps]))
                want.append(int((torch.tensor([[n]]).float() * frac).floor().clamp_min(1).long().item()))
            else:
                want.append(0)
        assert keep_schedule(n, steps) == want


def test_image_processor_restatement_round_trip():
    """utils/image_utils.py pil_to_unit_tensor / unit_tensor_to_pil = VaeImageProcessor(vae_scale_factor, do_normalize=False)
    .preprocess / .postprocess(output_type="pil") as the reference calls them (utils/image_utils.py:39,73,165-166)."""
    import numpy as np
    from PIL import Image

    from mmada_parallel_amd.utils import image_utils as iu

    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(48, 80, 3), dtype=np.uint8)
    x = iu.pil_to_unit_tensor(Image.fromarray(arr), 16)
    assert x.shape == (1, 3, 48, 80) and x.dtype == torch.float32
    assert np.array_equal(np.asarray(iu.unit_tensor_to_pil(x)[0]), arr)          # same size: no resampling, exact round trip
    y = iu.pil_to_unit_tensor(Image.fromarray(arr[:45, :70]), 16)                # 70 x 45 -> 64 x 32 (multiples of 16)
    assert y.shape == (1, 3, 32, 64) and 0.0 <= float(y.min()) <= float(y.max()) <= 1.0
    toks = iu.add_break_line(list(range(6)), 2, 3, new_number=126084)
    assert toks == [0, 1, 2, 126084, 3, 4, 5, 126084]


def test_paint_and_break_token_layout_on_a_fake_tokenizer():
    """encode_img_with_breaks / encode_img_with_paint (reference utils/image_utils.py:159-173,175-284): special tokens, row
    breaks, VQ offset, and which latent cells an in- / out-painting rectangle masks (area down-sampling, > 0.5)."""
    import numpy as np
    from PIL import Image

    from mmada_parallel_amd.utils import image_utils as iu

    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, size=(32, 64, 3), dtype=np.uint8)          # H = 32, W = 64 -> 16 x 32 latent cells
    from helpers import FakeVq

    img, vq = Image.fromarray(arr), FakeVq()
    toks = iu.encode_img_with_breaks(img, vq, vae_scale_factor=2)
    assert toks[0] == 126349 and toks[-1] == 126350 and len(toks) == 2 + 16 * 33
    body = toks[1:-1]
    assert all(body[r * 33 + 32] == 126084 for r in range(16))
    x = torch.from_numpy(arr.astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    want = (torch.nn.functional.avg_pool2d(x[:, :1], 2).reshape(-1) * 63).round().long() + 126356
    assert [t for i, t in enumerate(body) if i % 33 != 32] == want.tolist()
    # rectangle of half the height and a quarter of the width, centred: pixel rows 8..23, columns 24..39 -> cells 4..11 x 12..19
    inp, vis = iu.encode_img_with_paint(img, vq, mask_h_ratio=0.5, mask_w_ratio=0.25, mask_mode="inpainting")
    out, _ = iu.encode_img_with_paint(img, vq, mask_h_ratio=0.5, mask_w_ratio=0.25, mask_mode="outpainting")
    assert vis.size == (64, 32) and len(inp) == len(out) == 16 * 33
    grid_in = torch.tensor([t for i, t in enumerate(inp) if i % 33 != 32]).view(16, 32)
    grid_out = torch.tensor([t for i, t in enumerate(out) if i % 33 != 32]).view(16, 32)
    inside = torch.zeros(16, 32, dtype=torch.bool)
    inside[4:12, 12:20] = True
    assert torch.equal(grid_in == 126336, inside) and torch.equal(grid_out == 126336, ~inside)
    assert torch.equal(grid_in[~inside], want.view(16, 32)[~inside]) and torch.equal(grid_out[inside], want.view(16, 32)[inside])
    assert np.asarray(vis)[16, 32].tolist() == [127, 127, 127] and np.asarray(vis)[0, 0].tolist() == arr[0, 0].tolist()
    # one cell of dilation grows the masked block by a ring
    dil, _ = iu.encode_img_with_paint(img, vq, mask_h_ratio=0.5, mask_w_ratio=0.25, dilate_latent_k=1)
    grid_d = torch.tensor([t for i, t in enumerate(dil) if i % 33 != 32]).view(16, 32)
    assert int((grid_d == 126336).sum()) == 10 * 10


def test_pixel_token_helpers_match_the_reference_functions():
    """tests/golden/image_utils_tokens.npz: the reference's own encode_img_with_breaks / encode_img_with_paint (imported with
    a stub `diffusers` whose VaeImageProcessor is the restatement in utils/image_utils.py) on a fake tokenizer and a seeded
    image — token layout, special ids, mask geometry for both modes, dilation and the nearest / bilinear mask samplers."""
    import numpy as np
    from PIL import Image

    from helpers import GOLDEN, FakeVq, PAINT_UTIL_CASES, paint_util_image
    from mmada_parallel_amd.utils import image_utils as iu

    z = np.load(os.path.join(GOLDEN, "image_utils_tokens.npz"))
    img, vq = Image.fromarray(paint_util_image()), FakeVq()
    assert iu.encode_img_with_breaks(img, vq, vae_scale_factor=2) == z["breaks"].tolist()
    for name, kw in PAINT_UTIL_CASES.items():
        toks, vis = iu.encode_img_with_paint(img, vq, **kw)
        assert toks == z[name + "_tokens"].tolist(), name
        assert np.array_equal(np.asarray(vis), z[name + "_vis"]), name
    codes = torch.arange(18 * 35).view(1, -1) % 64
    assert np.array_equal(np.asarray(iu.decode_vq_to_image(codes, None, None, 36, 70, vq)), z["decoded"])
    import pytest

    with pytest.raises(ValueError):
        iu.decode_vq_to_image(codes[:, :-1], None, None, 36, 70, vq)


def _gemm8_program(nk, na0, na1, nb, balanced=False):
    """Program order of ONE wave of csrc/gemm8.hip (Gemm8::run with Gemm8::tile, or Gemm8::tile_bal when `balanced`):
    ('stage', half, tile), ('read', half, tile), ('wait', n) and ('phase',) events.  Kept in step with the kernel by hand:
    it is the specification the kernel follows."""
    if balanced:
        return _gemm8_program_balanced(nk, na0, na1, nb)
    four = na0 + na1 + 2 * nb
    ev = [("wait", 0)]
    for n, t in (("A0", 0), ("B0", 0), ("B1", 0), ("A1", 0), ("A0", 1), ("B0", 1)):
        ev.append(("stage", n, t))
    ev += [("wait", four), ("phase",)]
    for T in range(nk):
        tail = 0 if T + 2 < nk else (1 if T == nk - 2 else 2)
        if tail < 2:
            ev.append(("stage", "B1", T + 1))
        ev += [("read", "A0", T), ("read", "B0", T), ("wait", four if tail < 2 else na1), ("phase",)]
        if tail < 2:
            ev.append(("stage", "A1", T + 1))
        ev += [("read", "B1", T), ("wait", four if tail < 2 else 0), ("phase",)]
        if tail == 0:
            ev.append(("stage", "A0", T + 2))
        ev += [("read", "A1", T), ("phase",)]
        if tail == 0:
            ev += [("stage", "B0", T + 2), ("wait", four)]
        elif tail == 1:
            ev.append(("wait", nb + na1))
        ev.append(("phase",))
    return ev


def _gemm8_program_balanced(nk, na0, na1, nb):
    """tile_bal: one half-tile staged, read and retired per phase; W half 0 of K-tile T+1 is read in P4 of K-tile T."""
    four = na0 + na1 + 2 * nb
    ev = [("wait", 0)]
    for n, t in (("A0", 0), ("B0", 0), ("B1", 0), ("A1", 0), ("B0", 1), ("A0", 1)):
        ev.append(("stage", n, t))
    ev += [("wait", four), ("phase",), ("read", "B0", 0)]
    for T in range(nk):
        tail = 0 if T + 2 < nk else (1 if T == nk - 2 else 2)
        if tail < 2:
            ev.append(("stage", "B1", T + 1))
        ev += [("read", "A0", T), ("wait", four if tail < 2 else na1), ("phase",)]
        if tail < 2:
            ev.append(("stage", "A1", T + 1))
        ev += [("read", "B1", T), ("wait", four if tail < 2 else 0), ("phase",)]
        if tail == 0:
            ev.append(("stage", "B0", T + 2))
        ev.append(("read", "A1", T))
        if tail < 2:
            ev.append(("wait", four if tail == 0 else na0 + nb + na1))
        ev.append(("phase",))
        if tail == 0:
            ev.append(("stage", "A0", T + 2))
        if tail < 2:
            ev += [("read", "B0", T + 1), ("wait", four if tail == 0 else nb + na1)]
        ev.append(("phase",))
    return ev


@pytest.mark.parametrize("balanced", [False, True])
@pytest.mark.parametrize("na0,na1,nb", [(2, 2, 2), (3, 3, 2), (2, 1, 2), (1, 1, 2), (3, 2, 1)])
def test_gemm8_schedule(na0, na1, nb, balanced):
    """The LDS-DMA queue of the 8-phase GEMM, simulated in program order for every piece split the kernel instantiates:
    a half-tile is read only after a counted wait of an EARLIER phase retired all its pieces (RAW: the reader may be a
    wave of the other, one-barrier-late group), a slot is re-staged at least two phases after its last read (WAR), the
    slot holds the K-tile that is read, and the queue is empty at the end."""
    cnt = {"A0": na0, "A1": na1, "B0": nb, "B1": nb}
    for nk in (2, 4, 6, 8, 64, 192):
        fifo, landed, slot, last_read, ph = [], {}, {}, {}, 0
        for e in _gemm8_program(nk, na0, na1, nb, balanced):
            if e[0] == "phase":
                ph += 1
            elif e[0] == "stage":
                _, n, t = e
                assert t < nk
                lr = last_read.get((t & 1, n))
                assert lr is None or ph - lr >= 2, f"WAR: {n}{t} staged in phase {ph}, slot last read in phase {lr}"
                slot[(t & 1, n)] = t
                fifo += [(n, t)] * cnt[n]
            elif e[0] == "wait":
                while len(fifo) > e[1]:
                    landed[fifo.pop(0)] = ph   # a half-tile has landed when its LAST piece is retired
            else:
                _, n, t = e
                assert slot.get((t & 1, n)) == t, f"slot {n} holds K-tile {slot.get((t & 1, n))} when {t} is read"
                assert (n, t) not in fifo, f"RAW: {n}{t} still in flight in phase {ph}"
                assert ph - landed[(n, t)] >= 1, f"RAW: {n}{t} read in the phase of its wait"
                last_read[(t & 1, n)] = ph
        assert not fifo


def test_bench_self_launch_relays_the_line_and_the_exit_code(tmp_path):
    """`python bench.py --gpus N` as a plain command: bench.self_launch runs the script under torch.distributed.run with N ranks
    (gloo here), relays everything but rank 0's JSON line to stderr, prints that line LAST on stdout, and returns the
    launcher's exit code (non-zero when a rank fails or no line came back)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = tmp_path / "ranks_ok.py"
    ok.write_text(
        "import os, sys, json\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "print('noise from rank', os.environ['RANK'], flush=True)\n"
        "dist.barrier()\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'metric': 'm', 'n_gpus': dist.get_world_size(), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.barrier()\n"
        "print('late noise', flush=True)\n"
        "dist.destroy_process_group()\n")
    bad = tmp_path / "ranks_bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    drv = ("import sys; sys.path.insert(0, %r); import bench; "
           "sys.exit(bench.self_launch(2, script=sys.argv[1], argv=['--gpus', '2', '--steps', '1']))" % root)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", drv, str(ok)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-1500:]
    out_lines = p.stdout.strip().splitlines()
    assert len(out_lines) == 1 and "noise" in p.stderr
    import json

    line = json.loads(out_lines[-1])
    assert line["n_gpus"] == 2 and line["argv"] == ["--gpus", "2", "--steps", "1"]
    p = subprocess.run([sys.executable, "-c", drv, str(bad)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_gemm_planner_picks_the_measured_best_on_the_committed_sweep():
    """csrc/gemm.hip prices every tile configuration with a fitted cost model.  The committed sweep (profiles/
    r06_gemm8_sweep_final.txt, taken with the -DMMADA_TUNE build of the FINAL round-6 product sources — tools/build_tune.py, no fork
    of any kernel: 32 (shape, M) points of the TP = 1 / 2 / 4 / 8 projection shapes, all 8-phase configurations and two 16-wave ones
    measured side by side) is the evidence for its constants: on every point the planner's pick (mmada_gemm_plan — host arithmetic,
    no GPU) must be a configuration whose MEASURED rate is within 3 % of the best measured one."""
    import os

    from mmada_parallel_amd import abi

    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gateup": (24576, 4096), "down": (4096, 12288),
              "qkv2": (6144, 4096), "o2": (4096, 2048), "gu2": (12288, 4096), "dn2": (4096, 6144),
              "qkv4": (3072, 4096), "o4": (4096, 1024), "gu4": (6144, 4096), "dn4": (4096, 3072),
              "qkv8": (1536, 4096), "o8": (4096, 512), "gu8": (3072, 4096), "dn8": (4096, 1536)}
    code_of = {300: 0, 301: 1, 302: 2, 303: 3, 1320: 1320, 1256: 1256}
    lib = abi.lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cols, points = None, 0
    for ln in open(os.path.join(root, "profiles", "r06_gemm8_sweep_final.txt")):
        p = ln.split()
        if not p or ln.startswith("#"):
            continue
        if p[0] == "shape":
            cols = [int(x[1:]) for x in p[2:] if x.startswith("v")]
            continue
        if p[0] not in shapes:
            continue
        M, (N, K) = int(p[1]), shapes[p[0]]
        tf = {code_of[v]: float(x) for v, x in zip(cols, p[2:]) if v in code_of}
        pick = lib.mmada_gemm_plan(M, N, K)
        assert pick in tf, f"{p[0]} M={M}: the planner picks configuration {pick}, which the sweep did not measure"
        assert tf[pick] >= 0.97 * max(tf.values()), f"{p[0]} M={M}: pick {pick} at {tf[pick]} TF, best {max(tf.values())} TF ({tf})"
        points += 1
    assert points == 32
    assert lib.mmada_gemm_plan(2440, 4096, 100) == -1   # K is not a multiple of the K-tile


def test_attention_plan():
    """The launch plan of the attention kernel (csrc/attention.hip: attention_chunks through mmada_attention_plan, host arithmetic):
    a (batch, head) pair's 16-row query groups are cut into the fewest workgroups with the least estimated time
    rounds x (largest SIMD share x key tiles x 0.405 us + 5.7 us).  The property DESIGN §3.2 rests on: BASELINE configs[1]
    (L = 2438 -> 153 groups, 32 heads) runs ONE round of 256 workgroups at batch 1 with at most 5 groups per SIMD, and B rounds
    at batch B (configs[4]: 48 sequences)."""
    from mmada_parallel_amd import abi

    lib = abi.lib()

    def cost(pairs, groups, keys, c):
        n = -(-groups // c)
        return -(-pairs * c // 256) * (-(-n // 4) * -(-keys // 64) * 0.405 + 5.7)

    for B in (1, 2, 3, 16, 48):
        c = lib.mmada_attention_plan(32 * B, 153, 2438)
        assert c == 8, (B, c)                       # 8 workgroups per head: 19-20 groups each, 5 per SIMD
        assert -(-153 // c) == 20 and -(-32 * B * c // 256) == B
    assert lib.mmada_attention_plan(32, 104, 1654) in range(5, 9)     # configs[0]
    for pairs, groups, keys in ((1, 1, 1), (4, 153, 2438), (8, 7, 100), (16, 21, 333), (24, 63, 1000), (32, 83, 1313), (96, 153, 2438),
                                (64, 147, 2349), (32, 16, 2438)):
        c = lib.mmada_attention_plan(pairs, groups, keys)
        assert 1 <= c <= groups and -(-groups // c) <= 24, (pairs, groups, c)       # a workgroup holds at most 8 waves x 3 groups
        lo = -(-groups // 24)
        best = min(cost(pairs, groups, keys, k) for k in range(lo, groups + 1))
        assert abs(cost(pairs, groups, keys, c) - best) < 1e-6, (pairs, groups, keys, c, best)
        assert all(cost(pairs, groups, keys, k) > best + 1e-9 for k in range(lo, c)), "the smallest count that reaches it"
    assert lib.mmada_attention_plan(0, 5, 64) == -1 and lib.mmada_attention_plan(3, 0, 64) == -1
