import os, subprocess, sys
sys.path.insert(0, "/root/repo")
from meshclust2_amd import synth
root = "/root/repo"
seqs, hdrs = synth.families(61, 320, 1000, family=16)
fa = "/tmp/k9_probe.fa"
synth.write_fasta(fa, seqs, hdrs)
golden = os.path.join(root, "tests", "golden")
for arena in (30000, 36000, 42000, 48000, 60000):
    env = dict(os.environ, MSC_CLUSTER_CENTRE_ARENA=str(arena), MSC_CLUSTER_PROFILE="1")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(golden, "weights_k9_u8.txt"), "--id", "0.9", "--output", "/tmp/probe.clstr", "--sparse"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    log = r.stdout.decode(errors="replace")
    same = open("/tmp/probe.clstr", "rb").read() == open(os.path.join(golden, "k9_u8.clstr"), "rb").read() if r.returncode == 0 else None
    print(arena, r.returncode, [l for l in log.splitlines() if "centre store" in l or "Number of clusters:" in l], same)
