"""zerovox_amd -- MI355X-native ZeroVOX synthesis path (HIP kernels behind a C-ABI, Python host API)."""
