// The build id of the library: newsreclib_amd/_build.py hashes every translation unit with the files it includes and
// hands the result to THIS unit only, so that `nrl_build_id()` of a loaded library can be compared with the sources
// lying beside it (and a kernel edit never recompiles the other units just to change the id).
#include "../../include/newsreclib_amd.h"

#ifndef NRL_BUILD_ID_VALUE
#define NRL_BUILD_ID_VALUE "00000000000000000000000000000000"
#endif

extern "C" const char* nrl_build_id(void) {
  static const char id[] = "NRL_BUILD_ID=" NRL_BUILD_ID_VALUE;   // (the prefix lets _build.py find it without dlopen)
  return id + 13;
}
