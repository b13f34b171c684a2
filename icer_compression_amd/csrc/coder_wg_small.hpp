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

// A fourth instance with FOUR wavefronts per workgroup, icer::wg4 (round 4, last day): the list kernel of a LONE frame.  Its list
// is led by fifty mid-sparse level-1 units -- one chunk in ten not blank, scattered: chains of 4-6 ms in the two-wave instance
// and what the whole launch waits for (tools/list_trace.py).  Per wave-cycle the window coder is as efficient there as the
// pipeline, there are just too few waves on a unit: four waves take four chunks per window and the launch 6.07 instead of
// 6.38 ms; eight waves need so much LDS that they push the pipeline's workgroups off the compute units (8.7 ms;
// profiles/r04_logs/r04_zh_list_kernel_width.log).  Batches keep the one-wave instance.
#pragma push_macro("ICER_WG_WAVES")
#undef ICER_WG_WAVES
#define ICER_WG_WAVES 4
#define ICER_WG_NS wg4
#include "coder_wg_impl.hpp"
#undef ICER_WG_NS
#pragma pop_macro("ICER_WG_WAVES")
