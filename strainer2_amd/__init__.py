"""strainer2_amd -- MI355X-native k-mer scrub/count path of Strainer2.

The product is native: ``lib/libstrainer_kmer.so`` (C host layer + hand-written HIP kernels
for gfx950, C-ABI in ``include/strainer_kmer.h``) and the drop-in ``bin/kmer_scrub_count``
program.  This Python package is only a thin ctypes binding over that C-ABI, used by the
tests and by ``bench.py``.  There is no Python or CPU compute path: if the shared library is
missing, importing :mod:`strainer2_amd.native` raises.
"""
from .native import pack_stream  # noqa: F401
from .native import (  # noqa: F401
    SKError,
    KmerContext,
    KmerUnion,
    Keyset,
    lib,
    library_path,
    cli_path,
    decode_file,
    run_cli_inprocess,
)

__all__ = ["SKError", "KmerContext", "KmerUnion", "Keyset", "lib", "library_path", "cli_path", "decode_file",
           "run_cli_inprocess"]
