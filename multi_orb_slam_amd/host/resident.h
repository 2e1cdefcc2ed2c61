// resident.h -- internal to libmorb_host.so: what ORBextractor::operator() / ExtractBatch left in HBM, for ORBmatcher's frame
// upload.  The reference builds a Frame from the two extractor calls (src/Frame.cc:182-185) and searches it a moment later
// (src/Tracking.cc:1267): the 32-byte descriptor rows the matcher needs on the device are the rows the extractor wrote
// there, so they need not cross the bus again.  Nothing is taken on trust: an entry serves a descriptor matrix only if
//   * the extraction ran on the CALLING thread (threads are told apart by a token that is never reused, not by std::thread::id;
//     the next extraction of that extractor, which overwrites the device rows, orders itself behind the kernel that reads them:
//     note_reader / take_reader below, per publishing extractor), and
//   * the matrix holds, byte for byte, the rows the extractor handed out (memcmp against the extractor's host copy) --
//     pointer identity, frame ids or sequence numbers are not consulted.
#pragma once
#include <cstdint>
#include <cstdlib>

namespace ORB_SLAM2 {
// the device every handle of the host classes is created on (MORB_DEVICE, default 0: one process per GPU)
inline int host_device() { static const int d = [] { const char* e = std::getenv("MORB_DEVICE"); return e ? std::atoi(e) : 0; }(); return d; }
namespace resident {

// `owner` identifies the publishing extractor (or batch slot); a second publish by the same owner replaces the first.
// host_rows / d_rows: n x 32 bytes on the host / on the device, both valid until the owner's next publish or retire.
void publish(const void* owner, const uint8_t* host_rows, const uint8_t* d_rows, int n);
void retire(const void* owner);
// device rows equal to the n x 32 bytes at `rows`, published by the calling thread; NULL if there are none.  *owner_out: who published them.
const uint8_t* find(const uint8_t* rows, int n, const void** owner_out = nullptr);
// The calling thread has enqueued device work on `stream` (a hipStream_t) that reads the rows `owner` published (find() served them).
// That owner's next extraction -- on the same thread: find() serves nobody else -- takes the stream with take_reader(owner) and
// orders itself behind that work on the device (orbx_wait_for_stream) before it overwrites the rows.  The stream is kept WITH THE
// ENTRY (one slot per thread was consumed by whichever extractor ran next, e.g. mpIniORBextractor for mpORBextractorLeft's
// rows); retire(owner) keeps it until it is taken.
void note_reader(const void* owner, void* stream);
void* take_reader(const void* owner);
// Everything the calling thread had enqueued on `stream` is known to have completed (a search on it returned its results): nobody
// reads served rows through it any more.
void reader_done(void* stream);
// inspection (tests / bench): lookups served from HBM / not served, on all threads
void stats(unsigned long* served, unsigned long* missed);

}  // namespace resident
}  // namespace ORB_SLAM2
