// HkKartAgent.cs — the ML-Agents surface of the reference's agents, forwarded to libhk.so.
//
// Replaces KartAgent / HierarchicalKartAgent on the karts of an HkRacingEnvController scene.  The controller steps E race
// instances on the GPU; the agents of ONE of them (DisplayEnv) are bound to Unity GameObjects so that ML-Agents (training or
// Heuristic play) and the scene's renderers keep working unchanged:
//   CollectObservations  <- this agent's slice of hk_get_observations        (HierarchicalKartAgent.CollectObservations, HKA:485-604)
//   OnActionReceived     -> the controller's action buffers -> hk_set_actions (KartAgent.OnActionReceived KA:440,
//                           HierarchicalKartAgent.InterpretDiscreteActions HKA:1371-1379: only LowMode == RL agents act on them)
//   Heuristic            keyboard -> (steer, {0 brake, 1 coast, 2 accelerate}), as KartAgent.Heuristic KA:508
// Rewards (KA:440-470, REC:174-237, 359-433) are computed on device when HkConfig.rewards = 1 and handed to ML-Agents with
// AddReward / the SimpleMultiAgentGroup from hk_get_rewards, once per Academy step.
// Source only: dotnet / Unity are not part of the build image.
using Unity.MLAgents;
using Unity.MLAgents.Actuators;
using Unity.MLAgents.Policies;
using Unity.MLAgents.Sensors;
using UnityEngine;
using KartGame.AI.Native;

namespace KartGame.AI
{
    public enum HkLowLevelMode { RL = HkConst.HK_LOW_RL, MPC = HkConst.HK_LOW_MPC, LQR = HkConst.HK_LOW_LQR }      // HKA:21-26
    public enum HkHighLevelMode { MCTS = HkConst.HK_HIGH_MCTS, Fixed = HkConst.HK_HIGH_FIXED }                  // HKA:28-33

    public class HkKartAgent : Agent
    {
        [Header("Wiring (as KartAgent.teamAgents / otherAgents, KA:63-65)")]
        public HkKartAgent[] teamAgents;
        public HkKartAgent[] otherAgents;
        [Header("Modes (HKA:62-64)")]
        public HkLowLevelMode LowMode = HkLowLevelMode.LQR;
        public HkHighLevelMode HighMode = HkHighLevelMode.Fixed;
        public AgentMode Mode = AgentMode.Inferencing;          // KA:34-38 (Training agents plan randomly and earn goal-timing rewards)
        [Header("gameParams (HKA:38-52)")]
        public int treeSearchDepth = 5;
        public int velocityBucketSize = 2;
        public int timePrecision = 100;
        public int sectionWindow = 2;

        [HideInInspector] public HkRacingEnvController envController;
        [HideInInspector] public int agentIndex;              // index in envController.Agents (= the agent axis of every hk_* array)

        protected override void Awake()
        {
            base.Awake();
            // HKA.Awake :413-426: Sensors.Length + sectionHorizon * 5 + 8 + 12 * (others + team) = hk_obs_dim
            var brain = GetComponent<BehaviorParameters>().BrainParameters;
            brain.VectorObservationSize = HkConst.HK_NUM_SENSORS + envController.sectionHorizon * 5 + 8 + 12 * (otherAgents.Length + teamAgents.Length);
        }

        public override void CollectObservations(VectorSensor sensor)
        {
            // the controller fetched hk_get_observations once for this Academy step; copy this agent's [obs_dim] slice in the
            // reference's order (8 own, 12 per teammate, 12 per opponent, 5 per horizon section, 9 ray distances)
            float[] obs = envController.Observations;
            int dim = envController.ObsDim;
            int off = (envController.DisplayEnv * envController.Agents.Length + agentIndex) * dim;
            for (int k = 0; k < dim; k++) sensor.AddObservation(obs[off + k]);
        }

        public override void OnActionReceived(ActionBuffers actions)
        {
            base.OnActionReceived(actions);
            // InterpretDiscreteActions (HKA:1371-1379): continuous[0] = steering, discrete[0] in {0 brake, 1 coast, 2 accelerate};
            // libhk latches them for LowMode == RL agents and ignores them otherwise (LQNG agents drive themselves)
            envController.SetAction(agentIndex, actions.ContinuousActions[0], actions.DiscreteActions[0]);
        }

        public override void Heuristic(in ActionBuffers actionsOut)
        {
            ActionSegment<float> continuousActions = actionsOut.ContinuousActions;
            ActionSegment<int> discreteActions = actionsOut.DiscreteActions;
            continuousActions[0] = Input.GetAxisRaw("Horizontal");
            bool acc = Input.GetButton("Accelerate"), brk = Input.GetButton("Brake");
            discreteActions[0] = (acc && brk) ? 1 : (acc ? 2 : (brk ? 0 : 1));
        }

        // called by the controller after every hk_step: pose of this kart in the displayed race instance
        public void ApplyState(in HkAgentState s, float kartY)
        {
            transform.position = new Vector3(s.px, kartY, s.pz);
            transform.rotation = Quaternion.Euler(0f, s.yaw * Mathf.Rad2Deg, 0f);
            gameObject.SetActive((s.flags & HkConst.HK_F_ENABLED) != 0);      // Deactivate(disable) KA:405-416
        }
    }
}
