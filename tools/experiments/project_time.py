import sys, json
sys.path.insert(0, "/root/repo")
import bench, multi_orb_slam_amd as m
for n, w, h in ((1000, 640, 480), (2000, 1280, 720)):
    r = bench.project_roofline(m, n, w, h, 200)
    print(n, w, h, r["avg_launch_us"], r["workload"])
