"""The entry point that allocates nothing (avsr_logmel_fwd_static): same results as avsr_logmel_fwd, bit for bit,
on two calls in a row that share the per-GPU scratch array, the second with a smaller input; then the same pair
again with the array filled with NaNs before each call (nothing an earlier call left there may be read)."""
import ctypes

import pytest
import torch

from mocov2_whisper_flamingo_amd import _lib, ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [ops.MEL_WHISPER, ops.MEL_SPEC_LN])
def test_logmel_static_matches_workspace_entry(variant):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    tables = ops.logmel_tables(variant, dev)
    g = torch.Generator().manual_seed(5)
    # first call: 4 clips of 30 s; second call: 2 clips of 7.3 s (fewer workgroups, fewer frames: most of the
    # scratch keeps the first call's values)
    poison = ctypes.CDLL(_lib.LIB_PATH).avsr_debug_logmel_static_poison      # exported, not in the declared interface
    poison.restype, poison.argtypes = ctypes.c_int32, [ctypes.c_void_p]
    for B, n, n_frames, poisoned in ((4, 480000, 3000, False), (2, 116800, 730, False),
                                     (4, 480000, 3000, True), (2, 116800, 730, True), (1, 48000, 300, True)):
        wave = (0.1 * torch.randn(B, n, generator=g)).to(dev)
        if poisoned:
            assert poison(torch.cuda.current_stream().cuda_stream) == 0
        ref = ops.logmel_fwd(wave, tables, variant, n_frames=n_frames, channels_last=True)
        out = torch.full((B, n_frames, 80), float("nan"), device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
        got = ops.logmel_fwd(wave, tables, variant, n_frames=n_frames, channels_last=True, out=out,
                             static_workspace=True)
        after = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
        assert after == before, "the static entry point must not allocate"
        assert got.data_ptr() == out.data_ptr()
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref)
    # a workspace beyond the fixed array is rejected with a clear error, not truncated
    B, n_frames = 16, 3000
    assert lib.avsr_logmel_workspace_floats(B, n_frames) > 2 * 1024 * 1024
    wave = torch.zeros(B, 480000, device=dev)
    with pytest.raises(_lib.AvsrHipError, match="unsupported"):
        ops.logmel_fwd(wave, tables, variant, n_frames=n_frames, channels_last=True, static_workspace=True)
