import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "mse_loss" in r[0]]
a, b = idx[-3], idx[-2]
win = rows[a:b]
t_mse = win[0][1]
sp = [r for r in win if "split_grad" in r[0]][0]
ad = [r for r in win if "adam_fused" in r[0]][-1]  # last optimizer kernel of the step
cc = [r for r in win if "concat_seq" in r[0]][0]
print("mse->split (head+cross bwd): %.0f us" % ((sp[1]-t_mse)/1e3))
print("split->end of last optimizer kernel (encoder bwd, wgrad drain, Adam): %.0f us" % ((ad[2]-sp[1])/1e3))
print("-> concat (encoders fwd of the next step): %.0f us" % ((cc[1]-ad[2])/1e3))
print("concat -> next mse (cross fwd + head): %.0f us" % ((win[-1][2]-cc[1])/1e3))
