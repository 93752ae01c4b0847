// resblock_pair16x3.inc instantiated for kernel size 7: see that file.
#define P3_KT 7
#include "resblock_pair16x3.inc"
