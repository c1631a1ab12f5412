// hdr2mip -- Radiance .hdr -> .mip (the reference's tools/hdr2mip): bin/hdr2mip in.hdr out.mip
#include <cstdio>
#include <cstring>
extern "C" int fj_hdr2mip(const char *hdr_path, const char *mip_path);
int main(int argc, char **argv)
{
  if (argc == 2 && std::strcmp(argv[1], "--help") == 0) { std::printf("Usage: hdr2mip inputfile(*.hdr, *.rgbe) outputfile(*.mip)\n"); return 0; }
  if (argc != 3) { std::fprintf(stderr, "error: invalid number of arguments.\nUsage: hdr2mip inputfile(*.hdr, *.rgbe) outputfile(*.mip)\n"); return -1; }
  return fj_hdr2mip(argv[1], argv[2]) ? -1 : 0;
}
