/*
 * pjd_pipeline.h -- C ABI of the pipelined batcher (libpjdpipe.so): files (or in-memory JPEGs) in,
 * BMP / RGB pictures out, with scan, H2D, GPU decode, D2H and the consumer of the pictures all
 * overlapping.
 *
 * It replaces the reference's producer/consumer pair
 *     mcu_prepare()   src/decoder_host.cpp:104-211   (producer thread: read_JPEG + Huffman + batching)
 *     offloading()    src/decoder_host.cpp:213-350   (consumer thread: DPU copy/exec/copy + write_BMP)
 *     main()          src/decoder_host.cpp:396-399   (the two std::threads and their queue)
 * with
 *     scan workers  -> batches of `batch_images` consecutive inputs
 *     GPU slots     -> each slot owns a pjd_ctx (its own HIP stream) and runs
 *                      create / upload / decode / packed download for one batch at a time, so
 *                      the H2D of one batch, the kernels of another and the D2H of a third overlap
 *     sink workers  -> hand every picture to the caller's sink (the CLI writes "<stem>.bmp")
 *
 * Several GPUs (the reference's dpu_alloc(DPU_ALLOCATE_ALL) + one picture per DPU, src/decoder_host.cpp:225,
 * 262-300): `devices` lists HIP ordinals; every device gets `slots` slots of its own and its own queue of
 * batches.  Batches are dealt to devices before scanning starts, longest-processing-time first on their input
 * bytes (pjd_pipe_assign) -- the CLI sorts inputs by size like the reference (:46-61), so consecutive batches
 * differ a lot in cost -- and a device that runs dry takes batches from the others' queues.  No picture ever
 * crosses from one GPU to another: the path has no exchange step.
 *
 * Inputs keep their order inside a batch; batches complete in any order.  Error behaviour is the
 * reference's: a file the scanner rejects produces its messages and no picture; a Huffman error
 * produces the message and the partial picture (src/decoder_host.cpp:120-123,181).
 */
#ifndef PJD_PIPELINE_H
#define PJD_PIPELINE_H

#include "pjd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Called once per input, from a sink worker thread (several may run at once):
 *   index   position of the input in the caller's list
 *   name    the path / name given for it
 *   log     what the reference would have printed while parsing this file ("" if nothing)
 *   status  -1: rejected by the scanner (no picture), -2: the GPU batch failed (no picture),
 *           else PJD_ST_* of the entropy decode (0 = clean; > 0: partial picture, message =
 *           pjd_status_string(status))
 *   data/len  the picture in the requested output format; valid only during the call          */
typedef void (*pjd_pipe_sink)(void *user, int index, const char *name, const char *log, int status,
                              const uint8_t *data, uint64_t len);

#define PJD_PIPE_MAX_DEVICES 16

typedef struct pjd_pipe_opts {
    int32_t device;          /* HIP device ordinal                                             */
    int32_t out_format;      /* PJD_OUT_BMP / PJD_OUT_RGB8                                     */
    int32_t batch_images;    /* inputs per GPU batch               (0 -> 1024)                 */
    int32_t scan_threads;    /* JPEG scanner workers               (0 -> 4)                    */
    int32_t slots;           /* GPU batches in flight              (0 -> 3)                    */
    int32_t sink_threads;    /* workers calling the sink           (0 -> 4)                    */
    pjd_pipe_sink sink;      /* may be NULL (pictures are dropped: measurement only)           */
    void *sink_user;
    const int32_t *devices;  /* HIP ordinals to spread the batches over; NULL -> { device }    */
    int32_t n_devices;       /* entries in `devices` (at most PJD_PIPE_MAX_DEVICES)            */
    uint32_t scan_options;   /* PJD_SCAN_* of pjd_host.h handed to the scanner (0 = the reference's accept / reject set) */
} pjd_pipe_opts;

typedef struct pjd_pipe_stats {
    double wall_s;           /* whole run                                                      */
    double scan_s, create_s, upload_s, exec_s, download_s, sink_s;   /* summed over workers   */
    uint64_t n_inputs, n_decoded, n_rejected, n_batches, n_batch_failures;
    uint64_t pixels, in_bytes, ecs_bytes, out_bytes;
    uint64_t n_devices;                              /* devices that opened                    */
    uint64_t n_stolen;                               /* batches run by another device than the one they were dealt to */
    uint64_t device_batches[PJD_PIPE_MAX_DEVICES];   /* batches run per entry of `devices` (an entry whose device did not open stays 0) */
    uint64_t device_in_bytes[PJD_PIPE_MAX_DEVICES];  /* input bytes of those batches           */
    uint64_t n_exact_images;                         /* pictures decoded by the exact one-lane kernel (routed up front or
                                                        re-decoded after the parallel decoder flagged them)                 */
} pjd_pipe_stats;

/* Returns PJD_OK, PJD_E_NODEVICE if no slot could open the device, PJD_E_ARG.                 */
int pjd_pipe_run_files(const char *const *paths, int n, const pjd_pipe_opts *opts, pjd_pipe_stats *stats);
int pjd_pipe_run_memory(const uint8_t *const *data, const uint64_t *len, const char *const *names, int n,
                        const pjd_pipe_opts *opts, pjd_pipe_stats *stats);

/* The dealing rule on its own (no GPU needed): item k of cost[k] goes to device_of[k] in [0, n_devices);
 * items are taken by descending cost (ties: lower index first) and given to the least loaded device
 * (ties: lower device first).  Returns PJD_OK or PJD_E_ARG.                                     */
int pjd_pipe_assign(const uint64_t *cost, int n, int n_devices, int32_t *device_of);

/* GPU slots (context, buffer pools, page-locked output buffer) are kept between runs; this frees them. */
void pjd_pipe_release(void);

#ifdef __cplusplus
}
#endif
#endif
