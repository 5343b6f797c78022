// Table of per-instance HBM fields (base pointer + bytes per instance) used by the generic
// instance-copy kernel and by snapshot/restore.
#pragma once
#include <stddef.h>
struct DrlgxField {
  char *base;
  size_t stride;  // bytes per instance (multiple of 4)
  int is_vm;      // virtual-map arrays are rebuilt by every step and not copied into rollouts
  int pad;
};
