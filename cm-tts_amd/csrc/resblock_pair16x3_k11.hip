// resblock_pair16x3.inc instantiated for kernel size 11: see that file.
#define P3_KT 11
#include "resblock_pair16x3.inc"
