/* orb_rt.h -- minimal runtime plumbing exported next to the extractor/matcher ABIs: device buffers, streams and
 * HIP-event timing for callers that are not HIP programs themselves (ctypes/cgo/JNI hosts, bench.py).
 * Nothing here has a counterpart in the reference (it is CPU-only); it exists so that a host can keep data
 * resident in HBM between orbx_* / orbm_* calls and time kernels on the stream they run on.
 */
#ifndef ORB_RT_H
#define ORB_RT_H
#include "orb_types.h"
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int orb_device_count(void);                       /* number of HIP devices (0 if none)                         */
int orb_device_name(int device, char* out, int cap); /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"             */
int orb_set_device(int device);
int orb_malloc(void** d_ptr, size_t bytes);       /* hipMalloc on the current device                           */
int orb_free(void* d_ptr);
int orb_malloc_host(void** h_ptr, size_t bytes);  /* page-locked host memory: H2D copies from it are true async DMA      */
int orb_free_host(void* h_ptr);
int orb_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream); /* async when stream != NULL   */
int orb_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int orb_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
int orb_memset(void* d_dst, int value, size_t bytes, void* stream);
int orb_stream_sync(void* stream);                /* NULL = default stream                                     */
int orb_device_sync(void);
int orb_event_create(void** ev);
int orb_event_destroy(void* ev);
int orb_event_record(void* ev, void* stream);
int orb_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop                  */

#ifdef __cplusplus
}
#endif
#endif
