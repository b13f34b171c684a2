// coder_wg_small.hpp -- a second instance of the workgroup-window coder (coder_wg.hpp) with TWO wavefronts per workgroup,
// icer::wgs.  It takes the coding units that are all but blank (route_units_kernel): they are a first small window and
// then runs of blank chunks closed by one wave (blank_run), so the sixteen waves and 87 KiB of LDS of the full instance
// would only keep other workgroups off the compute unit; this one needs 40 KiB and shares the compute unit with the
// pipeline's workgroups.
#pragma once
#include "coder_wg.hpp"

#pragma push_macro("ICER_WG_WAVES")
#undef ICER_WG_WAVES
#define ICER_WG_WAVES 2
#define ICER_WG_NS wgs
#include "coder_wg_impl.hpp"
#undef ICER_WG_NS
#pragma pop_macro("ICER_WG_WAVES")

// A third instance with ONE wavefront per workgroup, icer::wg1 (round 4).  The phase profile of the two-wave instance on the
// units that are 90-95 % blank (tools/wgs_phase_profile.py, profiles/r04_logs/r04_g_wgs_phase_profile.log) has half of the
// waves' time at barriers: a window there is one chunk with content and one blank chunk, the blank run in between is closed
// by one wave while the other waits, and every one of a window's eight barriers costs the pair its skew.  A single wave has
// no one to wait for: a window is one chunk, the blank chunk behind it joins the next closed-form run.
#pragma push_macro("ICER_WG_WAVES")
#undef ICER_WG_WAVES
#define ICER_WG_WAVES 1
#define ICER_WG_NS wg1
#include "coder_wg_impl.hpp"
#undef ICER_WG_NS
#pragma pop_macro("ICER_WG_WAVES")
