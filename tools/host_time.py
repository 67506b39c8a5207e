import sys, os, time, subprocess, tempfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from segalign_amd import synth
from segalign_amd.build import build_host
import test_gpu_host as T
t_recs = T.make_records(100, [40000, 15000, 60000, 9000, 30000])
q_recs = []
for i, (name, s) in enumerate(t_recs[::-1]):
    m = synth.mutate(s, 200 + i, 0.09, indel_every=700)
    q_recs.append(("q%d" % (i + 1), m))
d = tempfile.mkdtemp()
T.write_fasta(os.path.join(d, 't.fa'), t_recs); T.write_fasta(os.path.join(d, 'q.fa'), q_recs, width=70)
exe = build_host()
for chunk in (20000, 2500):
    out = os.path.join(d, 'o%d' % chunk); os.mkdir(out)
    cmd = [exe, os.path.join(d, 't.fa'), os.path.join(d, 'q.fa'), './', '--wga_chunk=%d' % chunk, '--lastz_interval=45000', '--seq_block_size=1000000', '--outdir=' + out, '--num_threads=3', '--num_gpu=1', '--debug']
    t0 = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, SEGALIGN_AMD_DEBUG='1')); dt = time.time() - t0
    print('chunk', chunk, 'rc', r.returncode, 'sec', round(dt, 2)); print(r.stderr.decode()[-1500:])
