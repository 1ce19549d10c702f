/* ORACLE (test infrastructure): unity build of the C restatement -- see oracle.h for scope and pin status. */
#include "group.c"
#include "scalar.c"
#include "msm.c"
#include "merlin.c"
#include "toolbox.c"
