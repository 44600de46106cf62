// Environment switches of libmellon_hip.so in one place.
//
// SUPPORTED (documented in INTEGRATION.md, read with std::getenv where they act):
//   MELLON_AMD_MIXED=0            pure-fp64 MAP solve (no 32-bit copy of the n x m buffer)          api_fit.hip
//   MELLON_AMD_MIXED_MIN_ELEMS    n m below which the 32-bit copy is not made (default 2^27)        api_fit.hip
//   MELLON_AMD_SURROGATE          float | fixed: format of the 32-bit copy                           api_fit.hip
//   MELLON_AMD_NN_PREFILTER=0     exact 1-NN without the fp16-split pre-filter; _MIN: its pair count cov_kernels.hip
//   MELLON_AMD_NO_CACHE=1         no caching allocator                                               alloc.hip
//   MELLON_AMD_RCCL               path of librccl                                                    comm.hip
//   MELLON_AMD_TRACE, MELLON_AMD_TIMING=0     diagnostics: stage prints; no per-evaluation events    api_*.hip
//   MELLON_AMD_EMULATE_RANKS=N    tools/emulate_rank.py: one process does the work of rank 0 of N     api_precond.hip
//   (binding: MELLON_AMD_DEVICE, _COMM, _COMM_TIMEOUT, _PORT, _TOKEN, _SHARE_GPU, _FORCE_COMM -- mellon_amd/distributed.py)
//
// EXPERIMENT KNOBS -- everything read through mln_experiment(): the tunables the sweeps under tools/ and a few tests turn
// (solver constants, kernel variants, paths that were measured and rejected).  They are IGNORED unless
// MELLON_AMD_EXPERIMENTAL=1 is set as well: a stale variable in somebody's shell cannot change what the library computes.
// Each knob's meaning and default sits next to its mln_experiment() call; DESIGN.md S4 / S6 list the measurements.
#pragma once
#include <cstdlib>

inline const char* mln_experiment(const char* name) {
  static const bool on = [] { const char* e = std::getenv("MELLON_AMD_EXPERIMENTAL"); return e && std::atoi(e) != 0; }();
  return on ? std::getenv(name) : nullptr;
}
