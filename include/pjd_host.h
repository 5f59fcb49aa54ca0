/*
 * pjd_host.h -- C ABI of the host-side companions of the decode path (libpjdhost.so):
 * the JPEG container scanner that feeds pjd_image_desc, and the BMP emitter.
 *
 *   reference                                        this library
 *   -----------------------------------------------  ----------------------------
 *   Header *read_JPEG(const std::string&)            pjd_scan_file / pjd_scan_memory
 *       src/headers/jpeg.h:189, jpeg_scanner.cpp:345
 *   void write_BMP(metadata, mcus, dpu, filename)    pjd_write_file (the device already
 *       src/headers/bmp.h:6, bmp_writer.cpp:19          produced the file image, PJD_OUT_BMP)
 *                                                    pjd_rgb_to_bmp (host-side formatter)
 *
 * The scanner accepts and rejects the same files as the reference's, prints the same
 * messages (collected in a log instead of std::cout), fills the same fields -- and in
 * addition records where each restart segment starts inside the destuffed stream, which
 * the reference's scanner discards (jpeg_scanner.cpp:423).
 */
#ifndef PJD_HOST_H
#define PJD_HOST_H

#include "pjd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pjd_scanned pjd_scanned;

/* Both return 0 when the file is a decodable baseline JPEG, 1 when the scanner rejected it
 * (the log then ends with "<name>: Error - Invalid JPEG\n", decoder_host.cpp:120-123),
 * 2 when the file could not be opened.  *out is always set unless the return is 2.       */
int pjd_scan_memory(const uint8_t *data, uint64_t len, const char *name, pjd_scanned **out);
int pjd_scan_file(const char *path, pjd_scanned **out);

/* The same with options.  PJD_SCAN_PROGRESSIVE: a progressive (SOF2) file is parsed scan by scan instead of being rejected at
 * its first inter-scan marker as the reference does (jpeg_scanner.cpp:425-430); the descriptor then carries PJD_F_PROGRESSIVE
 * and the scan list (pjd_scan_desc).  Baseline files are unaffected.  Not reference behaviour: opt-in.                    */
#define PJD_SCAN_PROGRESSIVE 1u
int pjd_scan_memory_ex(const uint8_t *data, uint64_t len, const char *name, uint32_t options, pjd_scanned **out);
int pjd_scan_file_ex(const char *path, uint32_t options, pjd_scanned **out);

/* The descriptor is valid until pjd_scanned_free; desc->ecs / seg_offsets point into it. */
const pjd_image_desc *pjd_scanned_desc(const pjd_scanned *s);
const char *pjd_scanned_log(const pjd_scanned *s);      /* what the reference would print   */
int pjd_scanned_valid(const pjd_scanned *s);
void pjd_scanned_free(pjd_scanned *s);

/* Reference metadata vector u32[276] for this image (decoder_host.cpp:156-178), for callers
 * that drive pjd_exec_dpu_payload.                                                        */
void pjd_scanned_metadata(const pjd_scanned *s, uint32_t *m276);

/* Tight RGB8 -> the reference's BMP file image (bmp_writer.cpp:19-67).  `out` must hold
 * pjd_output_size(w, h, PJD_OUT_BMP) bytes.                                               */
void pjd_rgb_to_bmp(const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out);

/* Write a buffer to a file in one go; returns 0 or -1 (message as the reference:
 * "<file>: Error - Unable to create BMP file").                                           */
int pjd_write_file(const char *path, const uint8_t *data, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif
