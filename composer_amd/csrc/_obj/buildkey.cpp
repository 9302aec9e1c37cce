extern "C" const char* cmp_build_key(void) { return "f680d856cbac083a421f544ecb71f4ac8b6d7d6e6bc1ba2be906bf7714c5e9a9"; }
