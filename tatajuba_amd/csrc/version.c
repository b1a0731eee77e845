/* tjamd_version (): the library's name and a hash of the sources it was built from (csrc and include files, computed by
 * the Makefile into srchash.inc), so that whoever loads a libtatajuba_amd.so can tell which tree it came from:
 * tatajuba_amd/build.py computes the same hash over the tree and refuses a library that was built from another one. */
#include "srchash.inc"

const char *tjamd_version (void) { return "tatajuba_amd 0.3 (gfx950) src " TJ_SRC_HASH; }
const char *tjamd_source_hash (void) { return TJ_SRC_HASH; }
