// qmps_knobs.h - the ONLY place where libqmps_hip reads the environment.
//
// documented_switch: kernel-selection switches that are part of the library's documented behaviour (include/qmps_hip.h,
//   section "environment switches"): they select between two implementations of the SAME computation and exist so that the
//   test-suite can run both against the oracle.
// tuning_knob: thresholds and schedules that were swept while tuning (profiles/EXPERIMENTS.md).  They are compiled in only
//   with -DQMPS_DEBUG_KNOBS (make EXTRA=-DQMPS_DEBUG_KNOBS); the shipped library ignores them.
#pragma once
#include <stdlib.h>

namespace qmps {

inline const char* documented_switch(const char* name) { return getenv(name); }

inline const char* tuning_knob(const char* name) {
#ifdef QMPS_DEBUG_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

}  // namespace qmps
