// build_stamp.cpp -- carries the content hash of the tree the library was built from (restartsqp_amd/build.py compiles
// this file last, with -DRSQP_SRC_HASH="<sha256 of every source, header and flag>"): build() compares it with the hash of
// the tree it finds and rebuilds when they differ, so a prebuilt .so never silently outlives its sources. build.py reads
// the marker from the file's bytes (it never dlopens a library it may be about to replace).
#ifndef RSQP_SRC_HASH
#define RSQP_SRC_HASH "unstamped"
#endif
extern "C" {
__attribute__((used)) const char rsqp_build_hash_marker[] = "RSQP_SRC_HASH=" RSQP_SRC_HASH;
const char *rsqp_build_hash(void) { return rsqp_build_hash_marker + 14; }
}
