"""attribute SIGPROF samples (tools/host_profile_job.cpp) to the innermost av1_recon.cpp line of each inline chain"""
import collections, subprocess, sys
out = subprocess.run(["addr2line", "-e", "/tmp/hp/host_profile_job", "-a", "-i", "-C"], stdin=open("/tmp/hp/pcs.txt"), capture_output=True, text=True).stdout
cnt = collections.Counter(); cur = None; total = 0
chains = []
for line in out.splitlines():
    if line.startswith("0x"):
        chains.append([])
    else:
        chains[-1].append(line.split(" (discriminator")[0])
for ch in chains:
    total += 1
    pick = next((l for l in ch if "av1_recon.cpp" in l), ch[-1] if ch else "?")
    cnt[pick.split("/")[-1]] += 1
print("samples", total)
for k, v in cnt.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    print("%6d %5.1f%%  %s" % (v, 100.0 * v / total, k))
