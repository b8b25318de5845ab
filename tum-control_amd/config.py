"""
Problem constants of the reference's shipped configuration (data, restated from the YAML/CSV
files the reference loads at construction):
  Config/EDGAR/veh_params_pred.yaml:3-25, Config/EDGAR/pacejka_params.yaml:3-12,
  Config/EDGAR/ggv.csv, Config/EDGAR/MPC_params.yaml:10-43, Config/EDGAR/sim_main_params.yaml:39-42,
  Prediction_Models/pred_model_dynamic_stm_pacejka.py:38-46 (g, rolling-resistance factors).
`load_reference_config()` reads the same values from a reference-style Config/ directory when
one is available (drop-in use inside a TUM-CONTROL checkout).
"""
import copy
import os

VEH = dict(lf=1.484, lr=1.644, m=2520.0, Iz=13600.0, ro=1.225, S=2.9, Cd=0.35,
           acc_min=-3.5, acc_max=2.5, delta_f_min=-0.610865, delta_f_max=0.610865,
           delta_f_dot_min=-0.322, delta_f_dot_max=0.322, jerk_min=-8.0, jerk_max=6.0,
           lat_acc_min=-5.886, lat_acc_max=5.886, veh_length=4.973, veh_width=1.941)
TIRE = dict(Bf=10.0, Cf=1.3, Df=15591.427, Ef=0.97, Br=10.0, Cr=1.6, Dr=24629.523, Er=0.97, mu=1.0489)
PHYS = dict(g=9.81, fr0=0.009, fr1=0.002, fr4=0.0003)
GGV = dict(v=[0, 4, 8, 11.11, 12, 20, 24, 28, 32, 37.5],
           ax=[3, 3, 3, 3, 2.5, 2.5, 2.5, 2.5, 2.5, 2.5],
           ay=[5.886] * 10)
MPC = dict(q_lon=2.8, q_lat=2.8, q_yaw=0.4, q_vel=0.2, r_jerk=38.1, r_steering_rate=101.4,
           s_lon=1.0, s_lat=1.0, s_yaw=1.0, s_vel=1.0, s_jerk=1.0, s_steering_rate=1.0,
           L1_pen=106.7, L2_pen=9.9, combined_acc_limits=2,
           stds=[0.0, 0.0, 0.0, 0.8, 0.35, 0.035, 0.0, 0.0], uncertainty_propagation_horizon=5,
           n_samples=10, gamma=0.8, expansion_degree=2)
SIM = dict(Ts=0.02, Tp=3.04, Ts_MPC=0.08, T=100.0,
           w_state_estimation=[0.15, 0.15, 0.01, 0.8, 0.35, 0.05, 0.005, 0.0],
           # Config/EDGAR/sim_main_params.yaml:44-80: the disturbance simulation of the closed-loop harness (both switched off in the
           # shipped file; magnitudes and distribution types as shipped)
           simulate_state_estimation=False, disturbance_type_state_estimation="gaussian",
           w_posx=0.15, w_posy=0.15, w_yaw=0.01, w_vlong=0.8, w_vlat=0.35, w_yawrate=0.05, w_delta_f=0.005,
           simulate_disturbances=False, disturbance_type_derivatives="uniform",
           w_posx_dot=0.8, w_posy_dot=0.8, w_yaw_dot=0.1, w_vlong_dot=1.1, w_vlat_dot=0.1, w_yawrate_dot=0.05, w_delta_f_dot=0.1)


def default_config():
    return copy.deepcopy(dict(veh=VEH, tire=TIRE, phys=PHYS, ggv=GGV, mpc=MPC, sim=SIM))


def load_reference_config(config_path, sim_main_params=None, mpc_params_file="EDGAR/MPC_params.yaml"):
    """Read the same numbers from a TUM-CONTROL style Config/ directory (yaml + ggv.csv)."""
    import csv
    import yaml
    cfg = default_config()
    if sim_main_params is None:
        with open(os.path.join(config_path, "EDGAR/sim_main_params.yaml")) as f:
            sim_main_params = yaml.safe_load(f)
    for k in ("Ts", "Tp", "Ts_MPC"):
        cfg["sim"][k] = sim_main_params[k]
    with open(os.path.join(config_path, sim_main_params["veh_params_file_MPC"])) as f:
        cfg["veh"].update({k: v for k, v in yaml.safe_load(f).items() if k in cfg["veh"]})
    with open(os.path.join(config_path, sim_main_params["tire_params_file_MPC"])) as f:
        t = yaml.safe_load(f)
    cfg["tire"].update(t["tire_params"]["front"]); cfg["tire"].update(t["tire_params"]["rear"]); cfg["tire"]["mu"] = t["mu"]
    with open(os.path.join(config_path, mpc_params_file)) as f:
        mp = yaml.safe_load(f)
    cfg["mpc"].update({k: v for k, v in mp.items() if k in cfg["mpc"]})
    # which OCP the controller classes build (NMPC_class.py:90-94): the NONLINEAR_LS formulation is the one that exists here;
    # the value is carried so that the controller mirrors can refuse anything else instead of silently assuming it
    cfg["mpc"]["costfunction_type"] = mp.get("costfunction_type", "NONLINEAR_LS")
    with open(os.path.join(config_path, mp["lookuptable_gg_limits"])) as f:
        rows = list(csv.DictReader(f))
    cfg["ggv"] = dict(v=[float(r["vel_max_mps"]) for r in rows], ax=[float(r["ax_max_mps2"]) for r in rows],
                      ay=[float(r["ay_max_mps2"]) for r in rows])
    return cfg
