// Test tooling: C entry point over the Delaunay triangulator that ships in the reference's libelas copy
// (elas/triangle.cpp, called as elas/elas.cpp:445-494 calls it).  Compiled together with that file from where it
// lies in the reference tree; nothing of it is copied here.
#include <cstdlib>
#include "elas/triangle.h"

extern "C" __attribute__((visibility("default")))
int ref_triangle_delaunay(const float *xy, int n, int *tri_out, int capacity) {
    struct triangulateio in, out;
    in.numberofpoints = n;
    in.pointlist = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < 2 * n; ++i) in.pointlist[i] = xy[i];
    in.numberofpointattributes = 0;
    in.pointattributelist = NULL;
    in.pointmarkerlist = NULL;
    in.numberofsegments = 0;
    in.numberofholes = 0;
    in.numberofregions = 0;
    in.regionlist = NULL;
    out.pointlist = NULL;
    out.pointattributelist = NULL;
    out.pointmarkerlist = NULL;
    out.trianglelist = NULL;
    out.triangleattributelist = NULL;
    out.neighborlist = NULL;
    out.segmentlist = NULL;
    out.segmentmarkerlist = NULL;
    out.edgelist = NULL;
    out.edgemarkerlist = NULL;
    char switches[] = "zQB";
    triangulate(switches, &in, &out, NULL);
    const int nt = out.numberoftriangles;
    for (int i = 0; i < 3 * nt && i < 3 * capacity; ++i) tri_out[i] = out.trianglelist[i];
    free(in.pointlist);
    free(out.pointlist);
    free(out.trianglelist);
    return nt;
}
