"""pollen_amd -- MI355X-native FlatGFA depth engine.

Only the node-depth / path-depth hot path of cucapra/pollen is here:
`flatgfa` is the host-side mirror of the reference interface over the C ABI in
include/flatgfa.h, `device` the device-level surface on caller-owned HBM buffers, `sharded`
the multi-GPU path sharding.  All depth arithmetic runs in hand-written HIP kernels
(pollen_amd/csrc/depth_device.hip); the package has no CPU fallback.
"""
from .flatgfa import (SHARD_NO_RCCL, SHARD_WHOLE_PATHS, FlatGFA, FlatGFAError, ShardedFlatGFA, device_count, format_float, load, parse, parse_bytes,
                      parse_stream_bytes, shard_cuts, synth, translate_prealloc)

__all__ = ["SHARD_NO_RCCL", "SHARD_WHOLE_PATHS", "ShardedFlatGFA", "FlatGFA", "FlatGFAError", "device_count", "format_float", "load", "parse", "parse_bytes",
           "parse_stream_bytes", "shard_cuts", "synth", "translate_prealloc"]
