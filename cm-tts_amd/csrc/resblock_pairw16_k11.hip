// resblock_pairw16.inc instantiated for kernel size 11 (bf16 and fp16): see that file.
#define PW16_KT 11
#include "resblock_pairw16.inc"
