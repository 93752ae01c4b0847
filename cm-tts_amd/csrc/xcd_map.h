// XCD-aware workgroup -> tile mapping for 2-D grids of (column tiles, utterances) on gfx950 (8 XCDs, each with its own 4-MB L2; workgroups are handed to the XCDs
// round robin in linear order, x fastest): with the plain mapping the eight neighbours of a run of column tiles sit on eight different XCDs, so the halo columns and
// the partly used cache lines two adjacent tiles share are fetched across the fabric once per XCD.  Here the workgroups an XCD receives — linear ids x, x + 8, ... —
// take CONSECUTIVE tiles: XCD x owns the contiguous range of tiles that starts at x q + min(x, r) (q = total / 8, r = total % 8: a bijection for every grid).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
    const unsigned gx = gridDim.x, total = gx * gridDim.y, lin = blockIdx.x + gx * blockIdx.y;
    const unsigned q = total >> 3, r = total & 7, x = lin & 7;
    const unsigned nl = x * q + (x < r ? x : r) + (lin >> 3);
    by = (int)(nl / gx);
    bx = (int)(nl - (unsigned)by * gx);
}
