"""kyber_amd -- MI355X-native batched group arithmetic behind dedis/kyber's
kyber.Point.Mul / pairing.Suite hot path.

The compute lives in ``lib/libkyberhip.so`` (hand-written HIP for gfx950, C ABI
declared in ``include/kyber_hip.h``).  This package is the host-side mirror of
the reference's interfaces for that path; it contains no arithmetic fallback:
if the HIP library is missing, importing ``kyber_amd._lib`` raises.
"""
__version__ = "0.1.0"


def release_stream(stream) -> None:
    """Free the per-stream device workspaces of a ``torch.cuda.Stream`` (or raw hipStream_t handle) that is about to
    be destroyed: kyb_stream_release (include/kyber_hip.h).  Long-lived streams never need this."""
    from ._lib import check, load

    handle = getattr(stream, "cuda_stream", stream)
    check(load().kyb_stream_release(handle), "kyb_stream_release")
