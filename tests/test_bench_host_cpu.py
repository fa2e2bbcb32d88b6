"""Host-side pieces of bench.py that need no GPU: the `single_stream` bracket around event-sampled steps restores every
second-stream switch it turned off (module flags and the environment variables read at call time), whatever they were."""
import os
import types

import bench


def test_single_stream_switches_off_and_restores_every_second_stream(monkeypatch):
    mod = types.SimpleNamespace(overlap_teacher=None, overlap_frozen_text=True)
    monkeypatch.delenv("DCLIP_BF16_WGRAD_STREAM", raising=False)
    monkeypatch.setenv("DCLIP_TEACHER_TEXT_STREAM", "1")
    with bench.single_stream(mod):
        assert mod.overlap_teacher is False and mod.overlap_frozen_text is False
        assert os.environ["DCLIP_BF16_WGRAD_STREAM"] == "0" and os.environ["DCLIP_TEACHER_TEXT_STREAM"] == "0"
    assert mod.overlap_teacher is None and mod.overlap_frozen_text is True
    assert "DCLIP_BF16_WGRAD_STREAM" not in os.environ and os.environ["DCLIP_TEACHER_TEXT_STREAM"] == "1"
    with bench.single_stream(mod, on=False):                      # a step that is not sampled: nothing is touched
        assert mod.overlap_frozen_text is True and "DCLIP_BF16_WGRAD_STREAM" not in os.environ
    try:
        with bench.single_stream(mod):
            raise RuntimeError("step failed")
    except RuntimeError:
        pass
    assert mod.overlap_teacher is None and mod.overlap_frozen_text is True and "DCLIP_BF16_WGRAD_STREAM" not in os.environ


def test_step_flops_and_teacher_flops_are_positive_and_scale():
    from dclip_amd import config as dcfg
    b32, l14 = dcfg.vit_b32(), dcfg.vit_l14()
    T = b32.text.max_position_embeddings
    assert bench.step_flops_per_image(b32, T) > 0
    one = bench.meta_teacher_flops(b32, b32, 256, 8, T)
    assert bench.meta_teacher_flops(b32, b32, 512, 8, T) == 2 * one
    assert bench.meta_teacher_flops(b32, l14, 256, 8, T) > 10 * one
