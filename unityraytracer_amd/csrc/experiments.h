// experiments.h — quarantine of the A/B, probe and diagnostic compile-time switches.
//
// The product is the library built by unityraytracer_amd/build.py with NO -DURT_* switch.  Every switch below changes what the
// kernels do or adds instrumentation, and exists only to MEASURE something on the GPU box (scripts/README.md).  The bodies of the
// round-3 A/B probes (URT_MINORITY, URT_VOTE_*, URT_EXTRA_NODE_LOADS, URT_PROBE_NOSTORE, URT_LDS_LEAF_SINGLE) left the kernels in
// round 4 — their results are in profiles/r03_logs/, their code in the history; later experiments are kept as patches next to
// their logs (profiles/r04_logs/*.patch) instead of as #ifdef bodies.  A translation unit that sees one of them must also see -DURT_EXPERIMENT, and a library built that way
// reports a NEGATIVE urt_abi_version(): unityraytracer_amd/_lib.py (and through it every test, bench.py and the smoke) refuses to
// load it unless the caller opts in with URT_ALLOW_EXPERIMENT=1 — which only the scripts under scripts/ do.
// unityraytracer_amd/build.py build_variant() adds -DURT_EXPERIMENT by itself.
#pragma once

#if defined(URT_STAMPS) || defined(URT_SCHED_OCC) || defined(URT_SERVE_OCC) || defined(URT_SERVE_SLEEP) || defined(URT_SKY_FASTWRAP) || \
    defined(URT_SAH_BINS) || defined(URT_AB)
#ifndef URT_EXPERIMENT
#error "A/B, probe and diagnostic switches (-DURT_STAMPS, -DURT_SCHED_OCC, -DURT_SAH_BINS, ...) need -DURT_EXPERIMENT: the library then reports a negative urt_abi_version and loaders refuse it unless they opt in (csrc/experiments.h)"
#endif
#endif

#ifdef URT_EXPERIMENT
#define URT_ABI_SIGN (-1)
#else
#define URT_ABI_SIGN 1
#endif
