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
