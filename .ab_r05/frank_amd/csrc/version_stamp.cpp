// The build stamp of libfrank_hip.so.  The Makefile compiles this file again whenever ANY object of the library has changed, so
// that two libraries that differ in a kernel never carry the same stamp (the stamp ties a profile under profiles/ to the binary it
// was taken from: tools/profile_r04.sh records it, bench.py prints the loaded library's beside the profile's).
extern "C" const char *fh_build_stamp(void) { return "frank_amd 0.4 (gfx950; built " __DATE__ " " __TIME__ ")"; }
