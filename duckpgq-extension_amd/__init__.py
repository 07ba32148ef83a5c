"""duckpgq-extension_amd — MI355X-native path finding for DuckPGQ (iterativelength / shortestpath /
cheapest_path_length over the in-memory CSR).

The product is two shared libraries built from `csrc/`:
  libpgq_hip.so  hand-written HIP kernels + the C-ABI drop-in boundary (include/pgq_hip.h)
  libpgq_udf.so  DuckDB-free host mirror of the reference's scalar-function layer (include/pgq_udf.h)
This package is only the ctypes binding used by tests/ and bench.py; there is no Python or CPU compute path:
every search call fails loudly if the HIP library or a GPU is missing.
"""
from .binding import (DeviceCSR, PgqError, PgqState, build_native, copy_bandwidth_gbps, get_stats, kclass_names,  # noqa: F401
                      lib_paths, load_hip, load_udf, reset_stats, set_option, get_option, get_default_option, init_devices)
