import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import dropin_leg
r = dropin_leg.run(640, 480, (1000, 1000), batch=False)
print(json.dumps(r))
r = dropin_leg.run(640, 480, (1000, 1000), batch=True)
print(json.dumps(r))
