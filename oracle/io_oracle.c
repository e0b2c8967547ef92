/* io_oracle.c -- TEST INFRASTRUCTURE: plain-C restatement of the two wire / disk formats either side of the path
 * (SURVEY 8(f) next-4), followed line by line so that tloam_amd/kitti_io.py can be held against them byte for byte:
 *
 *   orc_read_velodyne   readVelodyneToO3d   /root/reference/include/tloam/models/io/read_file.hpp:307-327
 *                       (Point4f() = (0, 0, 0, 1): read_file.hpp:89; HasNaNs: :145-147)
 *   orc_format_pose     FrontEnd::savePose  /root/reference/src/front_end/front_end.cpp:169-179
 *
 * The reference reads through a std::fstream; what matters of its semantics is restated with fread:
 *   - the loop runs while `readFile.good() && !readFile.eof()` (:315): the condition is tested BEFORE the two reads, so the
 *     iteration in which a read comes up short still finishes;
 *   - istream::read stores the characters it did get and sets eofbit | failbit when it got fewer than asked for; a read on
 *     a stream that is no longer good() extracts nothing;
 *   - every iteration starts from a fresh Point4f() = (0, 0, 0, 1) (:316) and appends it unless one of the four floats is
 *     NaN (:319) -- also the iteration that hit the end of the file: a file that ends on a record boundary therefore
 *     yields ONE extra point (0, 0, 0, intensity 1); a trailing partial record yields a point made of the bytes that
 *     were there and the defaults for the rest (and no extra point after it);
 *   - x, y, z and the intensity are widened to double (:320-321).
 * savePose writes the top three rows with `ofs << double` -- the stream defaults: precision 6, neither fixed nor
 * scientific, i.e. printf("%g") -- separated by single spaces, std::endl after element (2, 3) (:171-178).
 * Parity unpinned as everything here: the reference has no test for either function. */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

/* returns 0, or -1 if the file cannot be opened (the reference exits, :309-312); *n = points found; at most cap are stored */
int orc_read_velodyne(const char* path, double* xyz, double* intensity, size_t cap, size_t* n) {
  FILE* f = fopen(path, "rb");
  *n = 0;
  if (!f) return -1;
  int good = 1; /* readFile.good(): no eofbit, failbit or badbit */
  while (good) { /* `readFile.good() && !readFile.eof()`: eof implies !good */
    float p[4] = {0.f, 0.f, 0.f, 1.f}; /* Point4f point; */
    unsigned char* b = (unsigned char*)p;
    size_t got = fread(b, 1, 3 * sizeof(float), f); /* readFile.read((char*)&point.x(), 3 * sizeof(float)) */
    if (got < 3 * sizeof(float)) good = 0;
    if (good) { /* a read on a failed stream extracts nothing */
      got = fread(b + 3 * sizeof(float), 1, sizeof(float), f); /* readFile.read((char*)&point.intensity(), sizeof(float)) */
      if (got < sizeof(float)) good = 0;
    }
    if (!(isnan(p[0]) || isnan(p[1]) || isnan(p[2]) || isnan(p[3]))) { /* !point.HasNaNs() */
      if (*n < cap) {
        xyz[3 * *n + 0] = (double)p[0];
        xyz[3 * *n + 1] = (double)p[1];
        xyz[3 * *n + 2] = (double)p[2];
        intensity[*n] = (double)p[3];
      }
      ++*n;
    }
  }
  fclose(f);
  return 0;
}

/* pose: 4x4 ROW-major.  Writes the line savePose writes (with the trailing newline) into out; returns its length,
 * or -1 if cap is too small. */
int orc_format_pose(const double pose[16], char* out, size_t cap) {
  size_t len = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      char tmp[64];
      int m = snprintf(tmp, sizeof(tmp), "%g", pose[i * 4 + j]); /* ofs << pose(i, j) */
      if (m < 0 || len + (size_t)m + 2 > cap) return -1;
      memcpy(out + len, tmp, (size_t)m);
      len += (size_t)m;
      out[len++] = (i == 2 && j == 3) ? '\n' : ' ';
    }
  out[len] = '\0';
  return (int)len;
}
