// gather7.hip -- grid_pull / grid_grad / the grid gradient of the pull's backward for spline orders 6 and 7 through bricks of the image:
// gather5.hip compiled with 14^3-cell bricks and eight-slot rows (see its header).  Reference: interpol/nd.py:80-143, 216-288, splines.py:60-80.
#define IP_G5_HIGH
#include "gather5.hip"
