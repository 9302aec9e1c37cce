extern "C" const char* cmp_build_key(void) { return "9510d71cf8f297db403eef61bfc4f87d403858221a340bd78524e7170d66c7ae"; }
