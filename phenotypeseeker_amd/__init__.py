"""phenotypeseeker_amd -- MI355X-native k-mer association engine behind the PhenotypeSeeker
`modeling` / `prediction` commands.  Python host code over libpsk.so (hand-written HIP for
gfx950, C ABI in include/psk.h, bound with ctypes).  No CPU fallback: every stage raises
PskError when the library or the GPU is missing."""
__version__ = "0.1.0"
