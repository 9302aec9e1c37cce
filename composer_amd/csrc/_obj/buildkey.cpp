extern "C" const char* cmp_build_key(void) { return "832835f111dbfdf916f343089d7c077f9225c0ab2518ca891d0681a03a578ecc"; }
