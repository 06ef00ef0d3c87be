"""CPU restatement of highPerfStereo() (highPerfStereo/highPerfStereoLib.cpp:177-645) composed from the
oracle's routines.  Test infrastructure only.

The mesh is a parameter: Fade2D (the reference's triangulator) is a closed library, so the caller supplies
`triangulate(points [n,2] int32) -> [n_tri,3] int32` -- the tests pass the triangulation under test, which
is checked against the Delaunay definition separately (orc_delaunay_check)."""
import numpy as np

import oracle_lib as O


def high_perf_stereo(left, right, p, disparity_map, triangulate, band_radius=0):
    """left/right: [rows, cols] uint8; p: any object with StereoParameters' field names; disparity_map: [rows, cols]
    float32, updated in place.  Returns (points [n,2] int32, info dict)."""
    assert left.shape == right.shape == disparity_map.shape                     # .cpp:187-191
    L, R = O.crop16(left), O.crop16(right)                                      # .cpp:245-247
    rows, cols = L.shape
    if p.applyHistEqualization:                                                 # .cpp:249-253
        L, R = O.equalize_hist(L), O.equalize_hist(R)
    grad = O.laplacian_abs(L, p.kernelSize, float(p.scale), float(p.delta))     # .cpp:258-259
    pts = O.high_gradient_points(grad, p.gradThreshold)                         # .cpp:269-280
    # .cpp:289-292: GaussianBlur with Size((int)applyBlur, (int)applyBlur) = 1x1: the identity
    La, Ra = L, R
    if p.invertRows or p.traceLines:                                            # .cpp:296-335
        La = O.alter_rows(L, p.invertRows, p.nbRows, p.traceLines, p.nbLines, p.lineSize)
        Ra = O.alter_rows(R, p.invertRows, p.nbRows, p.traceLines, p.nbLines, p.lineSize)
    kl, _ = O.exfast_detect(La, p.minThreshold, p.adaptivity, nonmax=True)      # .cpp:358, :376
    kr, _ = O.exfast_detect(Ra, p.minThreshold, p.adaptivity, nonmax=False)     # .cpp:359, :383
    cl = O.census5x5(O.u8_to_s8(L))                                             # .cpp:377-378
    cr = O.census5x5(O.u8_to_s8(R))
    ml, mr, _ = O.sparse_match(cl, cr, kl.astype(np.float32), kr.astype(np.float32), p.maxDisp, 1.0,
                               p.uniqueness, p.leftRightStep)                   # .cpp:354-355, :390
    disparities = np.zeros((rows, cols), np.float32)                            # .cpp:435
    sup, seen = [], set()

    def insert(x, y):
        if (x, y) not in seen:
            seen.add((x, y))
            sup.append((x, y))

    for i in range(len(ml)):                                                    # .cpp:436-445
        if mr[i] < 0:
            continue
        x, y = int(kl[ml[i], 0]), int(kl[ml[i], 1])
        disparities[y, x] = np.float32(x) - np.float32(kr[mr[i], 0])
        insert(x, y)
    n_matches = len(sup)
    final_disp = np.ascontiguousarray(disparity_map[:rows, :cols])
    final_costs = np.full((rows, cols), 25, np.int8)                            # .cpp:448
    occ = 64                                                                    # .cpp:449
    for it in range(1, p.nIters + 1):                                           # .cpp:451
        if p.recordFullDisp and len(sup) >= 3:                                  # .cpp:454-466, :532-554
            sp = np.array(sup, np.int32)
            tri = triangulate(sp)
            if len(tri):
                txy = sp[tri].reshape(-1, 6)
                abc = np.zeros((len(tri), 3), np.float32)
                for t in range(len(tri)):
                    v = sp[tri[t]]
                    abc[t] = O.plane_fit([[v[k, 0], v[k, 1], disparities[v[k, 1], v[k, 0]]] for k in range(3)])
                O.plane_prior_raster(txy, abc, rows, cols, disparities)
        costs = np.full((rows, cols), p.tHigh, np.int8)                         # .cpp:573
        if band_radius > 0:                                                     # build extension: banded search around the prior
            O.cost_eval_band(cl, cr, pts, disparities, costs, band_radius, p.maxDisp)
        else:
            O.cost_eval(cl, cr, pts, disparities, costs)                        # .cpp:574
        g = O.refine(pts, disparities, costs, p.tLow, p.tHigh, occ, final_disp, final_costs)   # .cpp:575-576
        if it != p.nIters:                                                      # .cpp:616-621
            nxy, _ = O.support_resampling(cl, cr, g, 5, p.costAggrWindowSize, p.tLow, p.tHigh, p.maxDisp, disparities)
            for x, y in nxy:
                insert(int(x), int(y))
            occ = max(1, occ // 2)
    for x, y in sup:                                                            # .cpp:624-637
        final_disp[y, x] = np.float32(int(disparities[y, x]))
    disparity_map[:rows, :cols] = final_disp
    points = np.concatenate([pts.reshape(-1, 2), np.array(sup, np.int32).reshape(-1, 2)]).astype(np.int32)
    return points, dict(n_high_grad=len(pts), n_left=len(kl), n_right=len(kr), n_matches=n_matches, n_support=len(sup),
                        prior=disparities, final_costs=final_costs)
