/*
 * crt_grid_pipe.h -- uniform grid, any hit, triangle-only scenes: gridShadowIntersectTest (intersect.h:196-308) with lane refill
 * and the cell walk running AHEAD of the object tests.
 *
 * The plain loop (k_grid<1, COUNT, true>) keeps 64 rays per wave until the longest is done; shadow rays differ in length by orders
 * of magnitude (most are occluded within a few cells, the rest cross the whole grid), and its counters say 0.24 of the lanes are
 * active per VALU instruction (profiles/r03b_pmc_grid_anyhit_plain.txt).  The one-thing-per-iteration refill form (k_grid_refill)
 * fixes the lengths and loses more than it gains: an iteration costs the union of its branches and a ray needs one iteration per
 * object test AND per few cells.
 *
 * Here an iteration has two phases that every lane can use:
 *   1. up to CRT_GP_STEPS DDA steps: the lane's walker goes on through the cells -- empty ones by the distance field, as in the
 *      plain loop -- until it finds an occupied cell, whose item range it keeps (`next`) before stepping on as the reference
 *      does when every object of the cell misses;
 *   2. up to CRT_GP_TESTS object tests from the range being tested (`pend`), which takes over `next` when it runs out.
 * The walker is therefore one occupied cell ahead of the tests: speculative, and dropped the moment a test finds t < bound.
 * Cells are met in the reference's order with the reference's accumulated t_next values (grid_step is the same function), the
 * objects of a cell in list order, ranges one after the other -- so the first t < light_dist - eps found is the reference's
 * (intersect.h:276-289), and the counters are those of the plain loop (a range's cells are counted when its tests begin).
 *
 * EXPERIMENT FILE, not built into the library (measured slower than the plain loop: DESIGN.md (d) Round 3, small-scene walks).
 * It was hooked up by including it in crt_gpu.hip after crt_wide8.h and, in dispatch_shadow's CRT_ACCEL_GRID case for triangle-only
 * scenes, `return launch(s, k_grid_shadow_pipe<COUNT>, la, st, o);`.
 */
#pragma once
#include "crt_kernels.h"

#ifndef CRT_GP_STEPS
#define CRT_GP_STEPS 6
#endif
#ifndef CRT_GP_TESTS
#define CRT_GP_TESTS 2
#endif
#ifndef CRT_GP_REFILL_MIN
#define CRT_GP_REFILL_MIN 16
#endif

template <bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK) void k_grid_shadow_pipe(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;
    bool alive = false;
    unsigned ri = 0;
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ld = 0.0f, bound = 0.0f, t = CRT_TMAX;
    GridWalk w; w.hits = false; w.ix = w.iy = w.iz = 0; w.sx = w.sy = w.sz = 1; w.ex = w.ey = w.ez = 0;
    w.tx = w.ty = w.tz = 0.0f; w.dtx = w.dty = w.dtz = 0.0f;
    bool walker_done = true;            /* the walk has left the grid (or never entered it) */
    bool found = false;                 /* a test returned t < bound: the ray is done */
    bool next_valid = false;            /* the walker holds an occupied cell's range and waits for `pend` to run out */
    int pk = 0, pe = 0, nk = 0, ne = 0; /* cell_items[pk .. pe) being tested; [nk .. ne) found by the walker */
    unsigned skip_left = 0;             /* empty cells ahead that need no look */
    unsigned seg_nodes = 0, next_nodes = 0;   /* COUNT: cells walked since the last range was found / on the way to `next` */

    RayFeed<true> feed;
    feed.start(la);
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_GP_REFILL_MIN) {
            if (feed.take(la, idle, !alive, &ri, &ray, &ld)) {
                bound = ld - CRT_K_EPSILON;                                            /* intersect.h:276 */
                grid_setup(sc, ray, &w);
                walker_done = !w.hits;
                found = false; next_valid = false;
                pk = pe = nk = ne = 0; skip_left = 0u;
                seg_nodes = next_nodes = 0u;
                t = CRT_TMAX;
                alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;
        /* ---- phase 1: the walkers ---- */
        for (int s = 0; s < CRT_GP_STEPS; s++) {
            const bool want = alive && !found && !walker_done && !next_valid;
            if (__ballot(want) == 0ull) break;
            if (want) {
                if (!(w.ix != w.ex && w.iy != w.ey && w.iz != w.ez)) walker_done = true;
                else {
                    if (skip_left == 0u) {
                        const int cell = w.iy * sc.grid.nz * sc.grid.nx + w.iz * sc.grid.nx + w.ix;
                        skip_left = sc.cell_dist[cell];
                        if (skip_left == 0u) {
                            nk = sc.cell_start[cell]; ne = sc.cell_start[cell + 1];
                            next_valid = true;
                            if (COUNT) { next_nodes = seg_nodes + 1u; seg_nodes = 0u; }
                            grid_step(&w);                                             /* what follows a cell whose objects all miss */
                        }
                    }
                    if (skip_left != 0u) {
                        if (COUNT) seg_nodes++;
                        grid_step(&w);
                        skip_left--;
                    }
                }
            }
        }
        /* ---- phase 2: the tests ---- */
        for (int j = 0; j < CRT_GP_TESTS; j++) {
            if (alive && !found && pk == pe && next_valid) {
                pk = nk; pe = ne; next_valid = false;
                if (COUNT) c_nodes += next_nodes;
            }
            const bool has = alive && !found && pk < pe;
            if (__ballot(has) == 0ull) break;
            if (has) {
                DHit h;
                const float tt = dev_object<1>(sc, sc.cell_items[pk], ray, true, &h);
                pk++;
                if (COUNT) c_prims++;
                if (tt < bound) { t = tt; found = true; }
            }
        }
        if (alive && (found || (walker_done && pk == pe && !next_valid))) {
            if (!found) {
                if (COUNT) c_nodes += seg_nodes;
                t = linear_shadow<1, COUNT>(sc, 0, sc.nng, ray, bound, CRT_TMAX, &c_prims);   /* intersect.h:298-307 */
            }
            if (la.t_out) la.t_out[ri] = t;
            if (la.occ_out) la.occ_out[ri] = (t < ld - CRT_K_EPSILON) ? 1 : 0;
            if (COUNT) c_rays++;
            alive = false;
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, true);
}
