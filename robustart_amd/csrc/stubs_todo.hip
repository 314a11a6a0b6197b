// Placeholder launchers for kernel families not written yet: they fail loudly with
// RART_ERR_UNSUPPORTED (never a silent CPU fallback).  Each is replaced by a real
// translation unit as it lands.
#include "rart_common.h"
#ifndef RART_HAVE_RESAMPLE
int rart_launch_resample(int id, const RartCorruptArgs&) { rart_set_error("%s: HIP kernel not implemented yet", rart_corruption_name(id)); return RART_ERR_UNSUPPORTED; }
size_t rart_ws_resample(int, int, int, int, int) { return 0; }
#endif
#ifndef RART_HAVE_JPEG
int rart_launch_jpeg(const RartCorruptArgs&) { rart_set_error("jpeg_compression: HIP kernel not implemented yet"); return RART_ERR_UNSUPPORTED; }
size_t rart_ws_jpeg(int, int, int, int) { return 0; }
#endif
#ifndef RART_HAVE_STENCIL
int rart_launch_stencil(int id, const RartCorruptArgs&) { rart_set_error("%s: HIP kernel not implemented yet", rart_corruption_name(id)); return RART_ERR_UNSUPPORTED; }
size_t rart_ws_stencil(int, int, int, int, int) { return 0; }
#endif
#ifndef RART_HAVE_COMPOSITE
int rart_launch_composite(int id, const RartCorruptArgs&) { rart_set_error("%s: HIP kernel not implemented yet", rart_corruption_name(id)); return RART_ERR_UNSUPPORTED; }
size_t rart_ws_composite(int, int, int, int, int) { return 0; }
#endif
