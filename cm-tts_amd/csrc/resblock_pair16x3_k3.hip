// resblock_pair16x3.inc instantiated for kernel size 3: see that file.
#define P3_KT 3
#include "resblock_pair16x3.inc"
