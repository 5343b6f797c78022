// Table of per-instance HBM fields (base pointer + bytes per instance) used by the generic
// instance-copy kernel and by snapshot/restore.
#pragma once
#include <stddef.h>
struct DrlgxField {
  char *base;
  size_t stride;  // bytes per instance (multiple of 4)
  int cls;        // 0 = belief/simulator state, 1 = virtual-map planes (rebuilt every step), 2 = ground-truth landmarks
  // what of the slice is live: 0 = all of it; 1 / 2 / 3 = unit_bytes per pose / landmark / factor of the SOURCE instance
  // (its counters): the copy moves the live part only, so that a large capacity costs short trajectories nothing
  int unit;
  int unit_bytes;
  int pad;        // > 0: bytes of the slice this entry stands for (a sub-slice: base points at it, stride is the whole slice's); 0: all of it
};
